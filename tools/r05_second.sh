# round 5: the GPU suite again, k_emit_wg's cycles by phase (a -DHD_EMIT_STATS build made on the box), the latency trace
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=gpurun_out/r05_second
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -q -m gpu "$@" > $OUT/pytest.log 2>&1
rc=$?
tail -30 $OUT/pytest.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi          # killed: no further GPU step
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make EXTRA=-DHD_EMIT_STATS ../libhipdeflate.so > /dev/null 2>&1)
for k in fastq text; do timeout -k 10 120 python3 tools/exp_emit_wg_stats.py 6 $k 2>&1 | tail -1 | tee -a $OUT/emit_wg_stats.txt || exit 1; done
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make ../libhipdeflate.so > /dev/null 2>&1)
HOOK_TRACE_N=16 timeout -k 10 300 bash tools/lat_trace.sh 6 > $OUT/lat_trace.txt 2>&1 || { tail -20 $OUT/lat_trace.txt; exit 1; }
cat $OUT/lat_trace.txt | cut -c1-220
for T in 1 16; do
  HIPDEFLATE_HOOK_STATS=1 BGZF_METHOD=hip6 timeout -k 10 60 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 >> $OUT/hook.jsonl 2>> $OUT/hook_stats.txt || exit 1
done
cat $OUT/hook.jsonl $OUT/hook_stats.txt
