# round 5: the emit kernel beside the parse reads the records with loads that are coherent at the device's level themselves (sc1, global_load) instead of
# invalidating its XCD's L2 per block (and through flat_load): the tests of the scheme, then the kernel trace with three / no resident wavefronts and the old order
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_sc1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 500 -k "span_of_sub or one_at_a_time or beside or stalls" > $O/pytest.log 2>&1 || { tail -25 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
bash tools/r05_keep_trace.sh || exit 1
cp gpurun_out/r05_keep_trace/summary.txt $O/keep_trace_l6.txt
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'stalls', j['verified'].get('stalls'))"; }
: > $O/ab.txt
for L in 5 3; do
  timeout -k 10 150 python3 bench.py --level $L --data text --block-kib 1024 --no-cpu --steps 5 --warmup 1 --no-extra 2>$O/err.log | line migz_l${L}_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
for L in 6 5 3; do
  timeout -k 10 150 python3 bench.py --level $L --no-cpu --steps 5 --warmup 1 --no-extra 2>$O/err.log | line encode_l$L | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
