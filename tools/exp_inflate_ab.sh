# on the GPU box (its copy of the tree is scratch): A/B of the inflate kernel's experiment switches on ONE box (boxes
# differ by a few per cent), each variant = a rebuild with -DHD_INF_... and the three decode streams of bench_decode3.sh
set -e
cd ${GRAFT_REPO_ROOT:?run this through gpurun: it rebuilds the library}
OUT=${1:-gpurun_out/inf_ab}
mkdir -p $OUT
: > $OUT/ab.txt
for v in ${VARIANTS:-"-DHD_INF_POLICY=0,-DHD_INF_DEFER=0,-DHD_INF_PREFETCH=0" "-DHD_INF_POLICY=1,-DHD_INF_DEFER=0,-DHD_INF_PREFETCH=0" "-DHD_INF_POLICY=0,-DHD_INF_DEFER=1,-DHD_INF_PREFETCH=0" "-DHD_INF_POLICY=1,-DHD_INF_DEFER=1,-DHD_INF_PREFETCH=0" "-DHD_INF_POLICY=1,-DHD_INF_DEFER=1,-DHD_INF_PREFETCH=1"}; do
  touch 7bgzf_amd/csrc/hd_api.hip
  make -s -C 7bgzf_amd/csrc EXTRA="$(echo $v | tr ',' ' ')" > $OUT/build.log 2>&1
  echo "== $v  $(grep -A12 k_inflate 7bgzf_amd/csrc/hd_api.resources.log | grep -E ' VGPRs:' | head -1 | sed 's/.*remark: *//')" | tee -a $OUT/ab.txt
  bash tools/bench_decode3.sh 2>&1 | tee -a $OUT/ab.txt
done
touch 7bgzf_amd/csrc/hd_api.hip
make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
