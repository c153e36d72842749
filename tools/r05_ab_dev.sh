# A/B of two versions of hd_device.hpp (tools/_ab/dev_before.hpp, dev_after.hpp: scratch) on one box: the scans' DPP moves with bound_ctrl; the whole parity file on the second
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_ab_dev; mkdir -p $O; : > $O/ab.txt
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'ratio', j['config'].get('ratio'))"; }
for v in before after before after; do
  cp tools/_ab/dev_$v.hpp 7bgzf_amd/csrc/hd_device.hpp
  touch 7bgzf_amd/csrc/hd_api.hip
  make -s -C 7bgzf_amd/csrc > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
  echo "== $v" | tee -a $O/ab.txt
  if [ $v = after ] && [ ! -e $O/tested ]; then
    timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 600 > $O/pytest.log 2>&1 || { tail -25 $O/pytest.log; exit 1; }
    tail -1 $O/pytest.log | tee -a $O/ab.txt; touch $O/tested
  fi
  timeout -k 10 150 python3 bench.py --no-cpu --steps 8 --warmup 2 --no-extra 2>$O/err.log | line encode_l1 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 150 python3 bench.py --level 2 --no-cpu --steps 5 --warmup 1 --no-extra 2>$O/err.log | line encode_l2 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 150 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line migz_l6_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 150 python3 bench.py --mode decode --stream libdeflate6 --no-cpu --steps 5 --warmup 2 --no-extra 2>$O/err.log | line decode_libdeflate6 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
