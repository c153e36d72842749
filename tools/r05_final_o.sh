# round 5: the reference's CLI on the batch pipelines at 2 GiB (start-up is 0.3 s of every run: the rates at 512 MiB are half start-up)
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_o; mkdir -p $O
timeout -k 10 900 bash tools/e2e_cielbox.sh $O 2048 > $O/e2e.log 2>&1 || { tail -20 $O/e2e.log; exit 1; }
cat $O/e2e_cielbox.txt
