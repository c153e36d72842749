# round 5, a bound: what the symbol count of the workgroup levels' emit kernel costs the STEP now that the kernel runs beside the parse -- builds
# without the count's LDS atomics (1) and without the count pass (2); with no counts every DEFLATE block gets the static code: valid streams
# (bench.py's round trip holds), larger, timing only
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_nocount; mkdir -p $O; : > $O/ab.txt
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'ratio', j['config'].get('ratio'))"; }
for v in 0 1 2 0; do
  touch 7bgzf_amd/csrc/hd_api.hip
  make -s -C 7bgzf_amd/csrc EXTRA="-DHD_EXP_NO_COUNT=$v" > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
  echo "== HD_EXP_NO_COUNT=$v" | tee -a $O/ab.txt
  timeout -k 10 150 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line migz_l6_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  HIPDEFLATE_NO_BESIDE=1 timeout -k 10 150 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line migz_l6_text_old_order | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 150 python3 bench.py --level 6 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line encode_l6 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
touch 7bgzf_amd/csrc/hd_api.hip
make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
