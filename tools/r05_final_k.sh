# round 5: `cielbox_hip 7bgzf -d` on the library's streaming decoder (the patch's batched loop) -- its tests, then the reference's CLI end to end
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_k; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_cielbox_hip.py -q -m gpu -x --timeout 700 > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
timeout -k 10 600 bash tools/e2e_cielbox.sh $O 512 > $O/e2e.log 2>&1 || { tail -20 $O/e2e.log; exit 1; }
cat $O/e2e_cielbox.txt
