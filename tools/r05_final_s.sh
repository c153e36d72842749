# round 5: the hook curve and the file-to-file rates on the final library
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_s; mkdir -p $O
timeout -k 10 500 bash tools/hook_curve.sh $O > $O/hook_curve.log 2>&1 || tail -3 $O/hook_curve.log
grep -E '"threads": (8|16), "method": "(hip1|hip2|hip5|hip6|libdeflate1|libdeflate6)' $O/hook_curve.jsonl | cut -c1-160
timeout -k 10 400 bash tools/e2e_files.sh $O 4 > $O/e2e_files.log 2>&1 || tail -3 $O/e2e_files.log
cat $O/e2e_files.txt
