cd ${GRAFT_REPO_ROOT:?}
OUT=gpurun_out/r05_inflate3
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_cielbox_hip.py -q -m gpu -x -k "inflate or decode or hip_inflate or unpipe or roundtrip or cielbox or smoke or flush" > $OUT/pytest.log 2>&1
rc=$?
tail -4 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 tools/inflate_call_latency.py $OUT/inflate_call_latency.jsonl | cut -c1-200 || exit 1
bash tools/r05_pipe_stats.sh
