# round 5: the two kernels of the workgroup levels side by side -- issue priority of the parse / of the emit wavefronts (s_setprio) and how many
# emit wavefronts a CU keeps, A/B on one box (each variant = a rebuild); MiGz level 6 on text and BGZF level 6 on the FASTQ-like set
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_prio; mkdir -p $O; : > $O/ab.txt
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'stalls', j['verified'].get('stalls'))"; }
for v in ${VARIANTS:-"-DHD_BESIDE_KEEP=3" "-DHD_BESIDE_PARSE_PRIO=3" "-DHD_BESIDE_EMIT_PRIO=3" "-DHD_BESIDE_KEEP=2" "-DHD_BESIDE_KEEP=2,-DHD_BESIDE_PARSE_PRIO=3" "-DHD_BESIDE_KEEP=3"}; do
  touch 7bgzf_amd/csrc/hd_api.hip
  make -s -C 7bgzf_amd/csrc EXTRA="$(echo $v | tr ',' ' ')" > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
  echo "== $v" | tee -a $O/ab.txt
  timeout -k 10 150 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line migz_l6_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 150 python3 bench.py --level 6 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line encode_l6 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
touch 7bgzf_amd/csrc/hd_api.hip
make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
