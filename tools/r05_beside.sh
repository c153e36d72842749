# round 5: the workgroup levels' emit kernel BESIDE the parse (hd_deflate_wg.hpp WgBeside): a small verified run first (guarded),
# then the A/B on one box at full size, then the parity tests
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_beside; mkdir -p $O; : > $O/ab.txt
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'ratio', j['config'].get('ratio'), 'stalls', j['verified'].get('stalls'))"; }
timeout -k 10 90 python3 bench.py --level 6 --no-cpu --steps 2 --warmup 1 --gib 1 --no-extra 2>$O/err0.log | line small_bgzf_l6 | tee -a $O/ab.txt || { tail -5 $O/err0.log; exit 1; }
timeout -k 10 90 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 2 --warmup 1 --gib 2 --no-extra 2>$O/err0.log | line small_migz_l6 | tee -a $O/ab.txt || { tail -5 $O/err0.log; exit 1; }
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export HIPDEFLATE_NO_BESIDE=1; else unset HIPDEFLATE_NO_BESIDE; fi
  echo "== HIPDEFLATE_NO_BESIDE=${HIPDEFLATE_NO_BESIDE:-unset}" | tee -a $O/ab.txt
  timeout -k 10 120 python3 bench.py --level 6 --no-cpu --steps 3 --warmup 1 --no-extra 2>$O/err.log | line bgzf_l6 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 120 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 3 --warmup 1 --no-extra 2>$O/err.log | line migz_l6_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 120 python3 bench.py --level 3 --data text --block-kib 1024 --no-cpu --steps 3 --warmup 1 --no-extra 2>$O/err.log | line migz_l3_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
