set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout 300 -p no:cacheprovider > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
for cfg in "migz6 --data text --block-kib 1024" "bgzf6 " "bgzf6_text --data text"; do
  set -- $cfg; name=$1; shift
  python3 bench.py --steps 3 --warmup 1 --level 6 --no-cpu --no-extra "$@" > $O/$name.log 2>&1 || tail -5 $O/$name.log
  grep '^{' $O/$name.log | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('$name', j['value'], j['ms_per_step'], j['config']['ratio'], j['roofline']['kernel_ms_avg'])
"
done
