# round 5: the batch inflater with the five-instruction walk as the default: the decode tests and the wide inflate run (batch kernel)
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_inf_walk; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu -x --timeout 300 -k "inflate or decode or unpipe or roundtrip or flush or bench or smoke or cielbox" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python3 tools/big_fuzz_inflate.py 200 31 32 33 34 35 36 > $O/big_fuzz_inflate.log 2>&1; echo "rc=$?"; tail -2 $O/big_fuzz_inflate.log
