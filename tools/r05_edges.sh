# round 5: the parse's per-lane edge values (keyed lanes, room at a piece's end, window limit) decided by the scalar unit where they are uniform: A/B on one box
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_edges; mkdir -p $O; : > $O/ab.txt
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'ratio', j['config'].get('ratio'), 'stalls', j['verified'].get('stalls'))"; }
for v in 0 1 0 1; do
  touch 7bgzf_amd/csrc/hd_api.hip
  make -s -C 7bgzf_amd/csrc EXTRA="-DHD_WG_UNIFORM_EDGES=$v" > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
  echo "== HD_WG_UNIFORM_EDGES=$v" | tee -a $O/ab.txt
  if [ $v = 1 ] && [ ! -e $O/tested ]; then
    timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 600 -k "workgroup or beside or span_of or level" > $O/pytest.log 2>&1 || { tail -25 $O/pytest.log; exit 1; }
    tail -1 $O/pytest.log | tee -a $O/ab.txt; touch $O/tested
  fi
  timeout -k 10 150 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line migz_l6_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  HIPDEFLATE_NO_BESIDE=1 timeout -k 10 150 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line migz_l6_text_old_order | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 150 python3 bench.py --level 6 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line encode_l6 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
touch 7bgzf_amd/csrc/hd_api.hip
make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
