# round 5: the wide differential runs on the box -- latency mode in every launch shape, the workgroup levels, the per-call inflater
cd ${GRAFT_REPO_ROOT:?}
OUT=gpurun_out/r05_fuzz
mkdir -p $OUT
timeout -k 10 900 python3 tools/big_fuzz_lat.py 600 31 32 33 34 35 36 37 38 > $OUT/big_fuzz_lat.log 2>&1; echo "lat rc=$?"; tail -2 $OUT/big_fuzz_lat.log
for s in 77 78 79 80; do timeout -k 10 300 python3 tools/big_fuzz_wg.py $s >> $OUT/big_fuzz_wg.log 2>&1; echo "wg $s rc=$?"; done; grep BIG_FUZZ $OUT/big_fuzz_wg.log
HD_FUZZ_PER_CALL=1 timeout -k 10 600 python3 tools/big_fuzz_inflate.py 60 21 22 23 24 > $OUT/big_fuzz_inflate_percall.log 2>&1; echo "inflate per call rc=$?"; tail -2 $OUT/big_fuzz_inflate_percall.log
timeout -k 10 600 python3 tools/big_fuzz.py > $OUT/big_fuzz_encode.log 2>&1; echo "encode rc=$?"; tail -2 $OUT/big_fuzz_encode.log
