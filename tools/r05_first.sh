# round 5, first run on the box: the GPU suite, then where a latency-mode batch of the workgroup levels spends its time
# (kernel trace of hipdeflate_lat_run alone), then the hook at 8 / 16 callers.  usage: bash tools/r05_first.sh [pytest args]
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=gpurun_out/r05_first
mkdir -p $OUT
timeout -k 10 800 python -m pytest tests -x -q -m gpu "$@" > $OUT/pytest.log 2>&1
rc=$?
tail -25 $OUT/pytest.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi          # killed: no further GPU step
HOOK_TRACE_N=16 timeout -k 10 300 bash tools/lat_trace.sh 3 6 > $OUT/lat_trace.txt 2>&1 || { tail -20 $OUT/lat_trace.txt; exit 1; }
cat $OUT/lat_trace.txt | cut -c1-220
for T in 1 8 16; do
  for M in hip3 hip6; do
    HIPDEFLATE_HOOK_STATS=1 BGZF_METHOD=$M timeout -k 10 60 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 >> $OUT/hook.jsonl 2>> $OUT/hook_stats.txt || exit 1
  done
done
cat $OUT/hook.jsonl $OUT/hook_stats.txt
