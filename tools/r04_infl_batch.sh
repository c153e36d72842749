set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r04_inflb; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_cielbox_hip.py -x -q --timeout 300 -p no:cacheprovider > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python3 tools/inflate_call_latency.py $O/inflate_call_latency.jsonl > $O/inflate_call.log 2>&1 || tail -5 $O/inflate_call.log
cat $O/inflate_call_latency.jsonl | cut -c1-160
timeout -k 10 300 bash tools/e2e_cielbox.sh $O/cielbox 512 > $O/cielbox.log 2>&1 || tail -5 $O/cielbox.log
sed -n 5,7p $O/cielbox/e2e_cielbox.txt
