# round 5: the cielbox_hip tests again, build_code's phases inside k_emit_wg, the hook with more batches in flight
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=gpurun_out/r05_third
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_cielbox_hip.py -q -m gpu > $OUT/pytest.log 2>&1
rc=$?
tail -8 $OUT/pytest.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make EXTRA=-DHD_EMIT_STATS ../libhipdeflate.so > /dev/null 2>&1)
for k in fastq text; do timeout -k 10 120 python3 tools/exp_emit_wg_stats.py 6 $k 2>&1 | tail -1 | tee -a $OUT/emit_wg_stats.txt || exit 1; done
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make ../libhipdeflate.so > /dev/null 2>&1)
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
for F in 2 4 8; do
  for T in 8 16 32; do
    HIPDEFLATE_INFLIGHT=$F HIPDEFLATE_HOOK_STATS=1 BGZF_METHOD=hip6 timeout -k 10 60 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 2>> $OUT/hook_stats.txt | sed "s/^{/{\"inflight\": $F, /" >> $OUT/hook.jsonl || exit 1
  done
done
cat $OUT/hook.jsonl; cat $OUT/hook_stats.txt
