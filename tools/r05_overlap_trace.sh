cd ${GRAFT_REPO_ROOT:?}
export TMPDIR=/tmp
O=gpurun_out/r05_overlap; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 1 --warmup 1 --gib 4 > $O/kt.log 2>&1
python3 - $O <<'PY'
import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/kt/*/*_kernel_trace.csv'))[-1]
rows=[r for r in csv.DictReader(open(f)) if 'k_parse_wg' in r['Kernel_Name'] or 'k_deflate_dynamic' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows[-24:]:
    print('%-18s q%s start %9.3f ms  dur %8.3f ms  grid %s' % (r['Kernel_Name'][:18], r.get('Queue_Id'), (int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r.get('Grid_Size_X', r.get('Grid_Size'))))
PY
