cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=gpurun_out/r05_seventh
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -q -m gpu -x -k "latency or lat or hook or room or capacity or stall or hip_deflate or flush" > $OUT/pytest.log 2>&1
rc=$?
tail -5 $OUT/pytest.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/r05_stats.sh | cut -c1-1300
HOOK_TRACE_N=16 timeout -k 10 300 bash tools/lat_trace.sh 6 > $OUT/lat_trace.txt 2>&1 || { tail -20 $OUT/lat_trace.txt; exit 1; }
grep "k_\|lat_run" $OUT/lat_trace.txt | cut -c1-170
