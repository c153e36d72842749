set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_span; mkdir -p $O; : > $O/ab3.txt
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'stalls', j['verified'].get('stalls'))"; }
for rep in 1 2 3 4; do
  for L in 5 6 3; do
    timeout -k 10 150 python3 bench.py --level $L --data text --block-kib 1024 --no-cpu --steps 6 --warmup 1 --no-extra 2>$O/err.log | line migz_l${L}_text | tee -a $O/ab3.txt || { tail -3 $O/err.log; exit 1; }
  done
done
