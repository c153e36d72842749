cd ${GRAFT_REPO_ROOT:?}
OUT=gpurun_out/r05_ninth
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
rc=$?
tail -5 $OUT/pytest.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python3 tools/inflate_call_latency.py $OUT/inflate_call_latency.jsonl > /dev/null || exit 1
cat $OUT/inflate_call_latency.jsonl | cut -c1-180
timeout -k 10 300 bash tools/e2e_cielbox.sh $OUT 512 > $OUT/e2e.log 2>&1 || { tail -5 $OUT/e2e.log; exit 1; }
cat $OUT/e2e_cielbox.txt
