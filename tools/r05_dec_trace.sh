# round 5: cielbox_hip 7bgzf -d -@16 -- a kernel trace of the decode: how many streams a launch of the latency inflater carries
# and what the gaps between launches are
cd ${GRAFT_REPO_ROOT:?}
export TMPDIR=/tmp
O=gpurun_out/r05_dec_trace; mkdir -p $O
D=/dev/shm/hd_cb2; mkdir -p $D
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
open('$D/in.bin','wb').write(s.fastq_like(64<<20, seed=1234).tobytes())
"
./oracle/_ref/cielbox_ref 7bgzf -l6 -@16 < $D/in.bin > $D/ref6.bgz
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- ./oracle/_ref/cielbox_hip 7bgzf -d -@16 < $D/ref6.bgz > $D/back.bin 2> $O/kt.log || { tail -3 $O/kt.log; rm -rf $D; exit 1; }
python3 - $O <<'PY'
import csv,glob,sys,collections,statistics
fs=sorted(glob.glob(sys.argv[1]+'/kt/*/*_kernel_trace.csv'))
rows=[r for r in csv.DictReader(open(fs[-1])) if 'k_inflate_lat' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
g=collections.Counter(int(r.get('Grid_Size_X') or r.get('Grid_Size'))//192 for r in rows)
print('launches', len(rows), 'streams per launch', sorted(g.items()))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
gap=[(int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']))/1e3 for i in range(len(rows)-1)]
print('kernel us: median %.0f mean %.0f; gap between launches us: median %.0f mean %.0f' % (statistics.median(d), statistics.mean(d), statistics.median(gap), statistics.mean(gap)))
print('span of all launches %.1f ms' % ((int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp']))/1e6))
PY
rm -rf $D
