# round 5: k_parse_wg with its LDS passed at launch (no register padding: 95 -> 80 VGPRs): parity first, then the rates
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_dynlds; mkdir -p $O; : > $O/ab.txt
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 200 -k "twin or wg or workgroup or migz or stall or latency" > $O/pytest.log 2>&1 || { tail -5 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, kernel ms', j['roofline']['kernel_ms_avg'], 'ratio', j['config'].get('ratio'), 'stalls', j['verified'].get('stalls'))"; }
timeout -k 10 150 python3 bench.py --level 6 --no-cpu --steps 3 --warmup 1 --no-extra 2>$O/err.log | line bgzf_l6 | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
timeout -k 10 150 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 3 --warmup 1 --no-extra 2>$O/err.log | line migz_l6_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
timeout -k 10 150 python3 bench.py --level 3 --data text --block-kib 1024 --no-cpu --steps 3 --warmup 1 --no-extra 2>$O/err.log | line migz_l3_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
timeout -k 10 150 python3 bench.py --level 5 --data text --block-kib 1024 --no-cpu --steps 3 --warmup 1 --no-extra 2>$O/err.log | line migz_l5_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
