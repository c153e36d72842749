# round 5: the two-wavefront latency inflater (hd_inflate.hpp PIPE): parity, per-call latency, and the batch kernel's rate
# (its source was restructured around the same lambdas: must not move)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=gpurun_out/r05_inflate
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_cielbox_hip.py -q -m gpu -x -k "inflate or decode or hip_inflate or unpipe or roundtrip or cielbox or smoke or flush" > $OUT/pytest.log 2>&1
rc=$?
tail -8 $OUT/pytest.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 tools/inflate_call_latency.py $OUT/inflate_call_latency.jsonl | cut -c1-200 || exit 1
STEPS=5 timeout -k 10 600 bash tools/bench_decode3.sh | tee $OUT/decode3.txt
