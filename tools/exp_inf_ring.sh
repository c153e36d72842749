set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/exp_guard.sh
exp_guard 7bgzf_amd/csrc/hd_inflate.hpp
for cfg in "2048 10" "2048 9" "4096 9"; do
  set -- $cfg
  sed -i "s/constexpr uint32_t INF_RING    = [0-9]*;/constexpr uint32_t INF_RING    = $1;/; s/constexpr uint32_t INF_LT_BITS = [0-9]*;/constexpr uint32_t INF_LT_BITS = $2;/" 7bgzf_amd/csrc/hd_inflate.hpp
  make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
  echo "== ring $1 ltbits $2"
  for a in "--stream libdeflate6" "--level 1"; do
  python bench.py --steps 2 --warmup 1 --no-cpu --gib 8 --tile-mib 32 --mode decode $a 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('  ', j['config'].get('stream'), j['value'], 'GB/s kernel ms', j['roofline']['kernel_ms_avg'])"
  done
done
