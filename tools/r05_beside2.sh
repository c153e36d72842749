# round 5: the emit kernel beside the parse as shipped (default for BGZF-sized blocks): the new parity test first (guarded), the whole
# GPU suite, the wide runs of the workgroup levels, then the rates of levels 3 / 5 / 6 on BGZF blocks with and without it
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_beside2; mkdir -p $O; : > $O/ab.txt
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 150 -k "beside" > $O/pytest0.log 2>&1 || { tail -8 $O/pytest0.log; exit 1; }
tail -1 $O/pytest0.log
timeout -k 10 600 python -m pytest tests -q -m gpu -x --timeout 300 -p no:cacheprovider > $O/pytest.log 2>&1 || { tail -8 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'ratio', j['config'].get('ratio'), 'stalls', j['verified'].get('stalls'))"; }
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export HIPDEFLATE_NO_BESIDE=1; else unset HIPDEFLATE_NO_BESIDE; fi
  echo "== HIPDEFLATE_NO_BESIDE=${HIPDEFLATE_NO_BESIDE:-unset}" | tee -a $O/ab.txt
  for L in 3 5 6; do
    timeout -k 10 120 python3 bench.py --level $L --no-cpu --steps 3 --warmup 1 --no-extra 2>$O/err.log | line bgzf_l$L | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  done
  timeout -k 10 120 python3 bench.py --level 6 --data text --no-cpu --steps 3 --warmup 1 --no-extra 2>$O/err.log | line bgzf_text_l6 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
