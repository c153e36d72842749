# round 4: the workgroup parse -- parity subset, then the three level-6 benches
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q --timeout 240 -p no:cacheprovider -k "fuzz_blocks or ratio_envelope or long_blocks or corpus or incompressible or capacity" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -3 $O/parity.log
python3 bench.py --steps 3 --warmup 1 --level 6 --data text --block-kib 1024 --no-cpu --no-extra > $O/migz6.log 2>&1 || tail -20 $O/migz6.log
python3 bench.py --steps 3 --warmup 1 --level 6 --no-cpu --no-extra > $O/bgzf6.log 2>&1 || tail -20 $O/bgzf6.log
python3 bench.py --steps 3 --warmup 1 --level 6 --data text --no-cpu --no-extra > $O/bgzf6_text.log 2>&1 || tail -20 $O/bgzf6_text.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_migz6 -- python3 bench.py --steps 2 --warmup 1 --level 6 --data text --block-kib 1024 --no-cpu --no-extra > $O/kt_migz6.log 2>&1 || true
for f in migz6 bgzf6 bgzf6_text; do grep '^{' $O/$f.log | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('$f', j['value'], j['ms_per_step'], j['config']['ratio'], j['roofline']['kernel_ms_avg'], j.get('verified',{}).get('how','')[:40])
"; done
find $O/kt_migz6 -name "*kernel_stats.csv" | head -1 | xargs head -5 | cut -c1-160
