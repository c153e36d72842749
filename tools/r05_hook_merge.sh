# round 5: the hook's merge policy (a complete batch waits for the one on the device and for its callers) against the old one
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_hook_merge; mkdir -p $O
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
export HIPDEFLATE_HOOK_STATS=1
for M in hip6 hip3 hip2 hip1; do
  for T in 4 8 16 32 64; do
    for F in 2 1; do
      echo -n "$M T=$T merge_inflight=$F: " | tee -a $O/merge.txt
      BGZF_METHOD=$M HIPDEFLATE_MERGE_INFLIGHT=$F timeout -k 5 60 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 1.5 2>&1 | tr '\n' ' ' | sed 's/.*batches (\([0-9.]*\) blocks each).*window \([0-9.]*\),.*device \([0-9.]*\).*"GBps_in": \([0-9.]*\).*"us_per_call": \([0-9.]*\).*/blocks per batch \1, window \2, device \3 us, \4 GB\/s, \5 us per call/' | cut -c1-200 | tee -a $O/merge.txt
      echo | tee -a $O/merge.txt
    done
  done
done
