# round 5: k_inflate's lane-group passes with the source byte read as "LDS ring, or device memory where the source is far" in two typed loads instead of the
# compiler's one flat_load_ubyte from a selected address: A/B on one box, then the decode tests on the variant
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_inf_split; mkdir -p $O
VARIANTS="-DHD_INF_SPLIT_SRC=0 -DHD_INF_SPLIT_SRC=1 -DHD_INF_SPLIT_SRC=0 -DHD_INF_SPLIT_SRC=1" STEPS=6 timeout -k 10 900 bash tools/exp_inflate_ab.sh $O || exit 1
touch 7bgzf_amd/csrc/hd_api.hip
make -s -C 7bgzf_amd/csrc EXTRA="-DHD_INF_SPLIT_SRC=1" > /dev/null 2>&1
timeout -k 10 600 python -m pytest tests -q -m gpu -k "inflate or decode" -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
