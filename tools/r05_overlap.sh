# round 5: the workgroup levels' emit kernel beside the next chunk's parse (hd_deflate_wg.hpp WgOverlap): A/B on one box
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_overlap; mkdir -p $O
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'ratio', j['config'].get('ratio'))"; }
for v in 1 0; do
  if [ $v = 1 ]; then export HIPDEFLATE_NO_OVERLAP=1; else unset HIPDEFLATE_NO_OVERLAP; fi
  echo "== HIPDEFLATE_NO_OVERLAP=${HIPDEFLATE_NO_OVERLAP:-unset}" | tee -a $O/ab.txt
  timeout -k 10 200 python3 bench.py --level 6 --no-cpu --steps 3 --warmup 1 2>$O/err.log | line bgzf_l6 | tee -a $O/ab.txt || exit 1
  timeout -k 10 200 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 3 --warmup 1 2>$O/err.log | line migz_l6_text | tee -a $O/ab.txt || exit 1
  timeout -k 10 200 python3 bench.py --level 3 --data text --block-kib 1024 --no-cpu --steps 3 --warmup 1 2>$O/err.log | line migz_l3_text | tee -a $O/ab.txt || exit 1
done
unset HIPDEFLATE_NO_OVERLAP
timeout -k 10 600 python -m pytest tests -q -m gpu -x -k "parity or bench" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
