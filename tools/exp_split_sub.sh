# experiment: blocks per parse/emit launch pair of the level-2 split path (run on the GPU box)
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/exp_guard.sh
exp_guard 7bgzf_amd/csrc/hd_deflate_dynamic.hpp
for g in "$@"; do
  sed -i "s/constexpr uint32_t SPLIT_SUB_BATCH = [0-9]*;/constexpr uint32_t SPLIT_SUB_BATCH = ${g};/" 7bgzf_amd/csrc/hd_deflate_dynamic.hpp
  make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
  echo "== sub-batch $g"
  python bench.py --steps 3 --warmup 1 --no-cpu --level 2 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  ', j['value'], 'GB/s')"
done
