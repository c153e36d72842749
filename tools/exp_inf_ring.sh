# usage: bash tools/exp_inf_ring.sh  -- experiment: the inflate kernel's LDS output ring (INF_RING) against occupancy, same box
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/exp_guard.sh
exp_guard 7bgzf_amd/csrc/hd_inflate.hpp
for R in ${@:-2048 4096 8192}; do
  sed -i "s/constexpr uint32_t INF_RING    = [0-9]*;/constexpr uint32_t INF_RING    = $R;/; s/static_assert(sizeof(InfLds) == 6400,/static_assert(sizeof(InfLds) >= 6400,/" 7bgzf_amd/csrc/hd_inflate.hpp
  touch 7bgzf_amd/csrc/hd_api.hip
  make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
  echo "== ring $R: $(grep -A12 'Function Name: _ZN2hd9k_inflate' 7bgzf_amd/csrc/hd_api.resources.log | grep -E 'VGPRs:|LDS Size|Occupancy' | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | tr '\n' ' ')"
  for a in "--stream libdeflate6" "--stream zlib6" "--level 1"; do
    python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extra --mode decode $a 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('   $a', j['value'], 'GB/s kernel ms', j['roofline']['kernel_ms_avg'])"
  done
done
