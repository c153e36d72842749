# round 5: where the parse's extra time beside the emit kernel comes from -- kernel trace of config 5 with three / two / one / NO emit wavefronts
# kept per CU (the parse's BESIDE instantiation alone on the device when none stays) and in the old order (the static instantiation)
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_keep_trace; mkdir -p $O; : > $O/summary.txt
for keep in ${KEEPS:-3 0 1 2 old}; do
  rm -rf /tmp/kt_$keep
  if [ $keep = old ]; then export HIPDEFLATE_NO_BESIDE=1; K=3; else unset HIPDEFLATE_NO_BESIDE; K=$keep; fi
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$keep -o kt -- python3 $GRAFT_REPO_ROOT/tools/bench_keep.py $K --level ${LEVEL:-6} --data text --block-kib 1024 --no-cpu --steps 4 --warmup 1 --no-extra > $O/bench_$keep.log 2>&1) || { tail -5 $O/bench_$keep.log; exit 1; }
  echo "== keep $keep: $(python3 -c "import json,sys; j=json.loads([l for l in open('$O/bench_$keep.log') if l.startswith('{\"metric\"')][-1]); print(j['value'], 'GB/s', j['ms_per_step'], 'ms/step')")" | tee -a $O/summary.txt
  f=$(find /tmp/kt_$keep -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY' | tee -a $O/summary.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'hd::k_' in r['Name'] and ('parse' in r['Name'] or 'dynamic' in r['Name'] or 'gate' in r['Name']):
        print('   %-90s calls %5s avg %10.1f us  total %8.1f ms' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done
