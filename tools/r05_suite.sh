cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_suite; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -q -m gpu -x --timeout 300 -p no:cacheprovider > $O/pytest.log 2>&1
rc=$?
tail -6 $O/pytest.log
echo "pytest rc=$rc"
