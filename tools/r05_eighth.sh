# round 5: the whole GPU suite, then the boundary's numbers: hook curve (ours + the reference's), the reference's own CLI on the backend
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=gpurun_out/r05_eighth
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
rc=$?
tail -8 $OUT/pytest.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 bash tools/hook_curve.sh $OUT > $OUT/hook_curve.log 2>&1 || { tail -5 $OUT/hook_curve.log; exit 1; }
cat $OUT/hook_curve.jsonl | cut -c1-200
timeout -k 10 300 bash tools/e2e_cielbox.sh $OUT 512 > $OUT/e2e.log 2>&1 || { tail -5 $OUT/e2e.log; exit 1; }
cat $OUT/e2e_cielbox.txt
