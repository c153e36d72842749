# round 5: the parse of a lone block shared by four workgroups (a.wg_split): parity, the kernel trace, the hook
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=gpurun_out/r05_fifth
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_cielbox_hip.py -q -m gpu -x -k "latency or lat or hook or cielbox or room or capacity or stall or hip_deflate or smoke or selftest or flush" > $OUT/pytest.log 2>&1
rc=$?
tail -12 $OUT/pytest.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
HOOK_TRACE_N=16 timeout -k 10 300 bash tools/lat_trace.sh 6 > $OUT/lat_trace.txt 2>&1 || { tail -20 $OUT/lat_trace.txt; exit 1; }
cat $OUT/lat_trace.txt | cut -c1-220
HOOK_TRACE_N=1 timeout -k 10 300 bash tools/lat_trace.sh 6 > $OUT/lat_trace1.txt 2>&1 || { tail -20 $OUT/lat_trace1.txt; exit 1; }
grep "k_" $OUT/lat_trace1.txt | cut -c1-220
for T in 1 8 16 32; do
  HIPDEFLATE_HOOK_STATS=1 BGZF_METHOD=hip6 timeout -k 10 60 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 >> $OUT/hook.jsonl 2>> $OUT/hook_stats.txt || exit 1
done
cat $OUT/hook.jsonl $OUT/hook_stats.txt
