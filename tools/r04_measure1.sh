set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r04_m1; mkdir -p $O
python3 tools/inflate_call_latency.py $O/inflate_call_latency.jsonl > $O/inflate_call.log 2>&1 || tail -5 $O/inflate_call.log
echo inflate_call done
bash tools/hook_curve.sh $O/hook > $O/hook.log 2>&1 || tail -5 $O/hook.log
echo hook done
bash tools/e2e_files.sh $O/e2e 4 > $O/e2e.log 2>&1 || tail -5 $O/e2e.log
cat $O/e2e/e2e_files.txt
cat $O/inflate_call_latency.jsonl
