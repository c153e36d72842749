set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q --timeout 200 -p no:cacheprovider -k "fuzz_blocks or ratio_envelope or long_blocks or corpus or incompressible or capacity" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
for cfg in "migz6 --level 6 --data text --block-kib 1024" "bgzf6 --level 6" "bgzf6_text --level 6 --data text" "migz3 --level 3 --data text --block-kib 1024" "bgzf3 --level 3"; do
  set -- $cfg; name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-extra "$@" > $O/$name.log 2>&1 || tail -5 $O/$name.log
  grep '^{' $O/$name.log | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('$name', j['value'], j['ms_per_step'], j['config']['ratio'], j['roofline']['kernel_ms_avg'])
"
  find $O/kt_$name -name "*kernel_stats.csv" | head -1 | xargs head -4 | cut -c1-130 | grep -v inflate
done
