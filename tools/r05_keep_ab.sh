# round 5: two or three emit wavefronts kept per CU, on the final kernels (tools/bench_keep.py sets it through hipdeflate_test_beside)
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_keep_ab; mkdir -p $O; : > $O/ab.txt
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'stalls', j['verified'].get('stalls'))"; }
for keep in 3 2 3 2; do
  echo "== keep $keep" | tee -a $O/ab.txt
  timeout -k 10 150 python3 tools/bench_keep.py $keep --level 6 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line encode_l6 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 150 python3 tools/bench_keep.py $keep --level 5 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line encode_l5 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 150 python3 tools/bench_keep.py $keep --level 6 --data text --block-kib 1024 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line migz_l6_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
