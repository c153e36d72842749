set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -q -m gpu -x --timeout 500 -k "beside or span_of or two_processes or one_at_a_time or stalls or pipe or hd7bgzf or workgroup or level" > $O/pytest.log 2>&1 || { tail -25 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'stalls', j['verified'].get('stalls'))"; }
timeout -k 10 150 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line migz_l6_text || { tail -3 $O/err.log; exit 1; }
timeout -k 10 150 python3 bench.py --level 6 --data text --block-kib 1024 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line migz_l6_text || { tail -3 $O/err.log; exit 1; }
timeout -k 10 150 python3 bench.py --level 6 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line encode_l6 || { tail -3 $O/err.log; exit 1; }
timeout -k 10 150 python3 bench.py --level 3 --no-cpu --steps 4 --warmup 1 --no-extra 2>$O/err.log | line encode_l3 || { tail -3 $O/err.log; exit 1; }
