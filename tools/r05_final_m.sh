# round 5: the whole GPU suite on the final tree
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout 400 -p no:cacheprovider > $O/pytest.log 2>&1 || { tail -12 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
