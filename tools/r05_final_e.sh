# round 5, final set E (after the emit kernel went beside the parse): the whole GPU suite, then the bench data kinds in launches of
# 1,000+ BGZF blocks at every level (the beside path) against the twin, then the workgroup levels' wide run
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_e; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu -x --timeout 300 -p no:cacheprovider > $O/pytest.log 2>&1 || { tail -8 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
HD_FUZZ_BLOCKS=65280 timeout -k 10 900 python3 tools/big_fuzz_synth.py 64 201 202 > $O/big_fuzz_synth.log 2>&1 || { tail -5 $O/big_fuzz_synth.log; exit 1; }
tail -1 $O/big_fuzz_synth.log
for s in 77 78; do timeout -k 10 300 python3 tools/big_fuzz_wg.py $s >> $O/big_fuzz_wg.log 2>&1 || { tail -3 $O/big_fuzz_wg.log; exit 1; }; done; grep BIG_FUZZ $O/big_fuzz_wg.log
