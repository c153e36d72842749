# round 5: the emit kernel beside the parse ACROSS sub-batches (one resident kernel per launch, two record buffers): the parity tests with
# small sub-batches forced? -- no: the 70,000-block test crosses a sub-batch boundary; then the rates
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_span; mkdir -p $O; : > $O/ab.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 150 -k "beside or many_small or wg or workgroup or migz or twin" > $O/pytest0.log 2>&1 || { tail -8 $O/pytest0.log; exit 1; }
tail -1 $O/pytest0.log
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'ratio', j['config'].get('ratio'), 'stalls', j['verified'].get('stalls'))"; }
for rep in 1 2; do
  for L in 3 6; do
    timeout -k 10 120 python3 bench.py --level $L --no-cpu --steps 3 --warmup 1 --no-extra 2>$O/err.log | line bgzf_l$L | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  done
  timeout -k 10 120 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 3 --warmup 1 --no-extra 2>$O/err.log | line migz_l6_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
