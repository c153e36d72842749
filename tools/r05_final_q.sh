set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_q; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -q -m gpu -x --timeout 600 > $O/pytest.log 2>&1 || { tail -25 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ms/step', j['ms_per_step'], 'ratio', j['config'].get('ratio'))"; }
for i in 1 2; do
timeout -k 10 150 python3 bench.py --no-cpu --steps 8 --warmup 2 --no-extra 2>$O/err.log | line encode_l1 || { tail -3 $O/err.log; exit 1; }
timeout -k 10 150 python3 bench.py --mode decode --stream libdeflate6 --no-cpu --steps 5 --warmup 2 --no-extra 2>$O/err.log | line decode_libdeflate6 || { tail -3 $O/err.log; exit 1; }
done
