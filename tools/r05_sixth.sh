# round 5: emit kernel with the blocks' codes built side by side + staged input for the shared parse: parity, phases, trace, hook
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=gpurun_out/r05_sixth
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_cielbox_hip.py -q -m gpu -x -k "latency or lat or hook or cielbox or room or capacity or stall or hip_deflate or smoke or selftest or flush" > $OUT/pytest.log 2>&1
rc=$?
tail -12 $OUT/pytest.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make EXTRA=-DHD_EMIT_STATS ../libhipdeflate.so > /dev/null 2>&1)
for k in fastq text; do timeout -k 10 120 python3 tools/exp_emit_wg_stats.py 6 $k 2>&1 | tail -1 | tee -a $OUT/emit_wg_stats.txt || exit 1; done
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make ../libhipdeflate.so > /dev/null 2>&1)
HOOK_TRACE_N=16 timeout -k 10 300 bash tools/lat_trace.sh 6 > $OUT/lat_trace.txt 2>&1 || { tail -20 $OUT/lat_trace.txt; exit 1; }
cat $OUT/lat_trace.txt | cut -c1-220
for T in 1 8 16 32 64; do
  HIPDEFLATE_HOOK_STATS=1 BGZF_METHOD=hip6 timeout -k 10 60 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 >> $OUT/hook.jsonl 2>> $OUT/hook_stats.txt || exit 1
done
BGZF_METHOD=hip3 timeout -k 10 60 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin 16 2 >> $OUT/hook.jsonl
cat $OUT/hook.jsonl $OUT/hook_stats.txt
