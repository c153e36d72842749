# round 4: parity of every level after the ladder moved (levels 3..5 = the workgroup parse with fewer ways), then the ladder timed
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r04_levels; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -x -q --timeout 300 -p no:cacheprovider > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
for L in 2 3 4 5 6; do
  for cfg in "bgzf_fastq " "bgzf_text --data text" "migz_text --data text --block-kib 1024"; do
    set -- $cfg; name=$1; shift
    python3 bench.py --steps 3 --warmup 1 --level $L --no-cpu --no-extra "$@" > $O/L${L}_$name.log 2>&1 || tail -5 $O/L${L}_$name.log
    grep '^{' $O/L${L}_$name.log | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('level $L $name', j['value'], 'GB/s', j['ms_per_step'], 'ms ratio', j['config']['ratio'])
"
  done
done | tee $O/ladder.txt
