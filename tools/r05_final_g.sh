# round 5: the workgroup levels where kernels run one at a time (launch-blocking runtime, a profiler collecting counters): the new test,
# the beside test, then the traffic counters of config 5 and encode_l6 under rocprofv3 --pmc (which dispatches one kernel at a time)
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
export TMPDIR=/tmp
O=gpurun_out/r05_final_g; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 400 -k "one_at_a_time or beside" > $O/pytest.log 2>&1 || { tail -15 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for spec in "migz_l6_text --level 6 --data text --block-kib 1024" "encode_l6 --level 6"; do
  set -- $spec; name=$1; shift
  timeout -k 10 300 bash tools/traffic_pmc.sh $name "$@" > $O/traffic_$name.log 2>&1 || { tail -5 $O/traffic_$name.log; tail -5 gpurun_out/traffic/${name}_FETCH_SIZE.log; exit 1; }
  tail -1 $O/traffic_$name.log | cut -c1-300
done
