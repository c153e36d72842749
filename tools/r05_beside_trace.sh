set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
export TMPDIR=/tmp
O=gpurun_out/r05_beside; mkdir -p $O
rm -rf $O/kt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 3 --warmup 1 --no-extra > $O/kt.log 2>&1 || { tail -3 $O/kt.log; exit 1; }
grep '^{' $O/kt.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', j['ms_per_step'])"
python3 - $O <<'PY'
import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/kt/*/*_kernel_trace.csv'))[-1]
rows=[r for r in csv.DictReader(open(f)) if 'k_parse_wg' in r['Kernel_Name'] or 'k_deflate_dynamic' in r['Kernel_Name'] or 'k_gate' in r['Kernel_Name'] or 'k_compact' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    n='parse' if 'parse' in r['Kernel_Name'] else 'gate' if 'gate' in r['Kernel_Name'] else 'compact' if 'compact' in r['Kernel_Name'] else 'emit'
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    if n in ('gate',) and d < 0.5: continue
    print('%-7s q%s start %9.2f end %9.2f dur %8.2f ms grid %s' % (n, r.get('Queue_Id'), (int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-t0)/1e6, d, r.get('Grid_Size_X', r.get('Grid_Size'))))
PY
