set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_span; mkdir -p $O; : > $O/ab2.txt
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'ratio', j['config'].get('ratio'), 'stalls', j['verified'].get('stalls'))"; }
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 150 -k "migz" > $O/pytest1.log 2>&1 || { tail -8 $O/pytest1.log; exit 1; }
tail -1 $O/pytest1.log
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export HIPDEFLATE_NO_BESIDE=1; else unset HIPDEFLATE_NO_BESIDE; fi
  echo "== HIPDEFLATE_NO_BESIDE=${HIPDEFLATE_NO_BESIDE:-unset}" | tee -a $O/ab2.txt
  for L in 6 5 3; do
    timeout -k 10 150 python3 bench.py --level $L --data text --block-kib 1024 --no-cpu --steps 3 --warmup 1 --no-extra 2>$O/err.log | line migz_l${L}_text | tee -a $O/ab2.txt || { tail -3 $O/err.log; exit 1; }
  done
done
