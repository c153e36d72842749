# round 5, final set F (the emit kernel beside the parse across sub-batches, 1 MiB members included): the whole GPU suite, the bench
# data kinds in launches of 1,000+ BGZF blocks at every level and of 520 one-MiB members at levels 3 / 6 against the twin
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_f; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu -x --timeout 300 -p no:cacheprovider > $O/pytest.log 2>&1 || { tail -8 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
HD_FUZZ_BLOCKS=65280 timeout -k 10 900 python3 tools/big_fuzz_synth.py 64 201 202 > $O/big_fuzz_synth.log 2>&1 || { tail -5 $O/big_fuzz_synth.log; exit 1; }
tail -1 $O/big_fuzz_synth.log
HD_FUZZ_BLOCKS=1048576 HD_FUZZ_LEVELS=3,6 timeout -k 10 900 python3 tools/big_fuzz_synth.py 520 301 > $O/big_fuzz_synth_migz.log 2>&1 || { tail -5 $O/big_fuzz_synth_migz.log; exit 1; }
tail -1 $O/big_fuzz_synth_migz.log
