# round 5, final set D: the whole GPU suite, then the wide differential runs, with the round's last kernels
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu -x --timeout 300 -p no:cacheprovider > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/r05_fuzz.sh
