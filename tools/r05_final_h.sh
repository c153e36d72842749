# round 5: the span of sub-batches made safe against a device with no resident emit wavefronts (a launch of the emit kernel in front of each
# gate) -- its test, the beside tests, then the A/B the change must not lose: config 5, level 5 / 3 on 1 MiB members, encode_l6, the default line
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_h; mkdir -p $O; : > $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 500 -k "span_of_sub or one_at_a_time or beside or stalls" > $O/pytest.log 2>&1 || { tail -25 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'stalls', j['verified'].get('stalls'))"; }
for rep in 1 2; do
  for L in 6 5 3; do
    timeout -k 10 150 python3 bench.py --level $L --data text --block-kib 1024 --no-cpu --steps 5 --warmup 1 --no-extra 2>$O/err.log | line migz_l${L}_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  done
  timeout -k 10 150 python3 bench.py --level 6 --no-cpu --steps 5 --warmup 1 --no-extra 2>$O/err.log | line encode_l6 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 150 python3 bench.py --level 2 --no-cpu --steps 5 --warmup 1 --no-extra 2>$O/err.log | line encode_l2 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
