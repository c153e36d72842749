# round 5: two processes on one card with resident emit wavefronts each; the stall test with members longer than the ring
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 700 -k "two_processes or stalls or span_of" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
