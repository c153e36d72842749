# usage: bash tools/exp_inf_waves7.sh  -- experiment, same box: the inflate kernel at SEVEN waves per SIMD (8-bit litlen and 7-bit offset
# direct tables: 4,864 B of LDS; amdgpu_waves_per_eu(7,7): 71 VGPRs + two spilled) against the shipped six
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/exp_guard.sh
exp_guard 7bgzf_amd/csrc/hd_inflate.hpp
cp 7bgzf_amd/csrc/hd_inflate.hpp /tmp/inf_orig.hpp
run() {
  touch 7bgzf_amd/csrc/hd_api.hip
  make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
  echo "== $1: $(grep -A14 'Function Name: _ZN2hd9k_inflate' 7bgzf_amd/csrc/hd_api.resources.log | grep -E 'VGPRs:|LDS Size|Occupancy|ScratchSize' | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | tr '\n' ' ')"
  for a in "--stream libdeflate6" "--stream zlib6" "--level 1"; do
    python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extra --mode decode $a 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('   $a', j['value'], 'GB/s kernel ms', j['roofline']['kernel_ms_avg'])"
  done
}
run "shipped (LT 9, DT 8)"
sed -i "s/constexpr uint32_t INF_LT_BITS = [0-9]*;/constexpr uint32_t INF_LT_BITS = 8;/; s/constexpr uint32_t INF_DT_BITS = [0-9]*;/constexpr uint32_t INF_DT_BITS = 7;/; s/static_assert(sizeof(InfLds) == 6400,/static_assert(sizeof(InfLds) <= 6400,/" 7bgzf_amd/csrc/hd_inflate.hpp
run "LT 8, DT 7, six waves"
sed -i "s/__global__ __launch_bounds__(64) void k_inflate/__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_inflate/" 7bgzf_amd/csrc/hd_inflate.hpp
run "LT 8, DT 7, seven waves"
cp /tmp/inf_orig.hpp 7bgzf_amd/csrc/hd_inflate.hpp
sed -i "s/constexpr uint32_t INF_DT_BITS = [0-9]*;/constexpr uint32_t INF_DT_BITS = 7;/; s/constexpr uint32_t INF_LT_BITS = [0-9]*;/constexpr uint32_t INF_LT_BITS = 8;/; s/constexpr uint32_t INF_RING    = [0-9]*;/constexpr uint32_t INF_RING    = 1024;/; s/static_assert(sizeof(InfLds) == 6400,/static_assert(sizeof(InfLds) <= 6400,/" 7bgzf_amd/csrc/hd_inflate.hpp
true
