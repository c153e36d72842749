# usage: bash tools/exp_params.sh "<WIN> <HASH>" ...   -- rebuild with other L1 geometry and bench (experiment only)
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/exp_guard.sh
exp_guard include/hipdeflate_params.h
mkdir -p gpurun_out
for cfg in "$@"; do
  set -- $cfg
  sed -i "s/#define HD_L1_WIN_BITS .*/#define HD_L1_WIN_BITS     $1/; s/#define HD_L1_HASH_BITS .*/#define HD_L1_HASH_BITS    $2/" include/hipdeflate_params.h
  make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
  echo "== WIN $1 HASH $2"
  python bench.py --steps 2 --warmup 1 --no-cpu --gib 8 --tile-mib 32 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(j['value'], 'GB/s ratio', j['config']['ratio'], 'kernel ms', j['roofline']['kernel_ms_avg'])"
done
