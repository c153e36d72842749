# usage: AB=<tag> FILE=<path of a header under 7bgzf_amd/csrc> bash tools/r05_ab_any.sh -- A/B of tools/_ab/<tag>_before.hpp / <tag>_after.hpp (scratch) as that header
# on one box: level 1 (the default line), level 2, decode; the whole parity file on the second version
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_ab_${AB:?}; mkdir -p $O; : > $O/ab.txt
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'ratio', j['config'].get('ratio'))"; }
for v in before after before after; do
  cp tools/_ab/${AB}_$v.hpp ${FILE:?}
  touch 7bgzf_amd/csrc/hd_api.hip
  make -s -C 7bgzf_amd/csrc > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
  echo "== $v" | tee -a $O/ab.txt
  if [ $v = after ] && [ ! -e $O/tested ]; then
    timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -q -m gpu -x --timeout 600 > $O/pytest.log 2>&1 || { tail -25 $O/pytest.log; exit 1; }
    tail -1 $O/pytest.log | tee -a $O/ab.txt; touch $O/tested
  fi
  timeout -k 10 150 python3 bench.py --no-cpu --steps 8 --warmup 2 --no-extra 2>$O/err.log | line encode_l1 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 150 python3 bench.py --data text --no-cpu --steps 8 --warmup 2 --no-extra 2>$O/err.log | line encode_l1_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 150 python3 bench.py --level 2 --no-cpu --steps 5 --warmup 1 --no-extra 2>$O/err.log | line encode_l2 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
