# experiment: scratch budget of the split path with 1 MiB MiGz blocks (run on the GPU box)
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/exp_guard.sh
exp_guard 7bgzf_amd/csrc/hd_deflate_dynamic.hpp
for g in "$@"; do
  sed -i "s/constexpr uint64_t SPLIT_SCRATCH_BUDGET = (uint64_t)[0-9]* << 20;/constexpr uint64_t SPLIT_SCRATCH_BUDGET = (uint64_t)${g} << 20;/" 7bgzf_amd/csrc/hd_deflate_dynamic.hpp
  make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
  echo "== budget $g MiB"
  for l in 2 3 6; do
  python bench.py --steps 2 --warmup 1 --no-cpu --level $l --data text --block-kib 1024 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  level $l', j['value'], 'GB/s', j['config']['ratio'])"
  done
done
