# on the GPU box (its copy of the tree is scratch): level 6's two-way table at 2560 buckets (8 parse waves per CU) against 1536
# (10 waves, the LDS footprint of level 5): speed and ratio on both BENCH data kinds.  (The twin is not rebuilt: bench.py only.)
set -e
cd ${GRAFT_REPO_ROOT:?run this through gpurun: it rebuilds the library}
for b in 2560 1536 2048; do
  touch 7bgzf_amd/csrc/hd_api.hip; make -s -C 7bgzf_amd/csrc EXTRA="-DHD_L6_BUCKETS=${b}u" > /dev/null 2>&1
  echo "== HD_L6_BUCKETS=$b: $(grep -A12 'k_deflate_staticILi13ELi12ELb1ELi5ELi1ELi1ELi1E' 7bgzf_amd/csrc/hd_api.resources.log | grep -E 'LDS Size' | head -1 | sed 's/.*remark: *//')"
  LEVELS=6 bash tools/bench_levels.sh | grep -v "level 9"
done
touch 7bgzf_amd/csrc/hd_api.hip; make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
