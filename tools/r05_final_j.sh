# round 5: the batch inflate kernel with the typed source loads (HD_INF_SPLIT_SRC) under the wide run of mutated streams
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_j; mkdir -p $O
timeout -k 10 900 python3 tools/big_fuzz_inflate.py 200 31 32 33 34 35 36 > $O/big_fuzz_inflate.log 2>&1 || { tail -5 $O/big_fuzz_inflate.log; exit 1; }
tail -2 $O/big_fuzz_inflate.log
