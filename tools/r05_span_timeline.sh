# round 5: the timeline of ONE step of config 5 (start / end of every kernel of the last step, ms from the step's first kernel): where the step's
# 127 ms go beyond four parses
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_span_timeline; mkdir -p $O
rm -rf /tmp/kt_tl
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_tl -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --level ${LEVEL:-6} --data text --block-kib 1024 --no-cpu --steps 3 --warmup 1 --no-extra > $O/bench.log 2>&1) || { tail -5 $O/bench.log; exit 1; }
f=$(find /tmp/kt_tl -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY' | tee $O/timeline.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'hd::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steps: a step starts with the memset-less first launch = k_deflate_dynamic<...,1> with grid 1536*64; take the last 3 resident launches
starts = [i for i, r in enumerate(rows) if 'k_deflate_dynamic' in r['Kernel_Name'] and int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0) in (1536 * 64, 1536)]
if len(starts) < 2:
    starts = [i for i, r in enumerate(rows) if 'k_gate' in r['Kernel_Name']][:1]
i0 = starts[-2] if len(starts) >= 2 else 0
i1 = starts[-1] if len(starts) >= 2 else len(rows)
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1]:
    if 'k_inflate' in r['Kernel_Name']:
        continue
    print('%-62s grid %8s  %9.3f -> %9.3f ms  (%8.3f)' % (r['Kernel_Name'].replace('void ', '')[:62], r.get('Grid_Size', r.get('Grid_Size_X', '?')), (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
PY
