set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
export TMPDIR=/tmp HIPDEFLATE_BESIDE=1
O=gpurun_out/r05_beside; mkdir -p $O
for cfg in "--level 6 --data text --block-kib 1024" "--level 6"; do
rm -rf $O/kt
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py $cfg --no-cpu --steps 1 --warmup 1 --no-extra > $O/kt.log 2>&1 || { tail -3 $O/kt.log; exit 1; }
python3 - $O "$cfg" <<'PY'
import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/kt/*/*_kernel_trace.csv'))[-1]
rows=[r for r in csv.DictReader(open(f)) if 'k_parse_wg' in r['Kernel_Name'] or 'k_deflate_dynamic' in r['Kernel_Name'] or 'k_gate' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[len(rows)//2:]
t0=int(rows[0]['Start_Timestamp'])
print('==', sys.argv[2])
for r in rows[:14]:
    n='parse' if 'parse' in r['Kernel_Name'] else 'gate' if 'gate' in r['Kernel_Name'] else 'emit'
    print('%-6s q%s start %9.3f ms  end %9.3f ms  dur %8.3f ms  grid %s' % (n, r.get('Queue_Id'), (int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r.get('Grid_Size_X', r.get('Grid_Size'))))
PY
done
