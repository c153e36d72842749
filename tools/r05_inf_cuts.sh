# round 5: the batch inflate kernel's cuts, A/B on one box (VARIANTS = the -D lists), then the inflate tests on the default build
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_inf_cuts; mkdir -p $O
VARIANTS="${VARIANTS:--DHD_INF_WALK5=0 -DHD_INF_WALK5=1 -DHD_INF_WALK5=0 -DHD_INF_WALK5=1}" STEPS=6 timeout -k 10 900 bash tools/exp_inflate_ab.sh $O || exit 1
cat $O/ab.txt
timeout -k 10 600 python -m pytest tests -q -m gpu -k "inflate or decode" -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
