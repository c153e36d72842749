# round 5: the latency inflater after the stale-tail fix and the wait guards: the long-codeword-run test alone first (guarded), then
# the decode tests, call latency, the per-call wide run
cd ${GRAFT_REPO_ROOT:?}
OUT=gpurun_out/r05_inflate6
mkdir -p $OUT
timeout -k 10 150 python -m pytest tests/test_gpu_boundary.py -q -m gpu -x --timeout 100 -k "long_codewords" > $OUT/pytest0.log 2>&1
rc=$?
tail -4 $OUT/pytest0.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -q -m gpu -x --timeout 120 -k "inflate or decode or hip_inflate or unpipe or roundtrip or flush" > $OUT/pytest.log 2>&1
rc=$?
tail -4 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 tools/inflate_call_latency.py $OUT/inflate_call_latency.jsonl | cut -c1-170 || exit 1
HD_FUZZ_PER_CALL=1 timeout -k 10 600 python3 tools/big_fuzz_inflate.py 60 21 22 23 24 > $OUT/big_fuzz_inflate_percall.log 2>&1; echo "inflate per call rc=$?"; tail -2 $OUT/big_fuzz_inflate_percall.log
