# round 5: the hook's batches in flight at levels 1 / 2 / 6, 8 and 16 callers (tools/exp_hook_inflight.sh for the workgroup levels)
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_hook_inflight; mkdir -p $O
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
export HIPDEFLATE_HOOK_STATS=1
for M in hip6 hip3 hip2 hip1; do
  for T in 8 16 32; do
    for F in 1 2 3; do
      echo -n "$M T=$T inflight=$F: " | tee -a $O/inflight.txt
      BGZF_METHOD=$M HIPDEFLATE_INFLIGHT=$F timeout -k 5 60 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 1.5 2>&1 | tr '\n' ' ' | sed 's/.*batches (\([0-9.]*\) blocks each).*device \([0-9.]*\).*"us_per_call": \([0-9.]*\).*/blocks per batch \1, device \2 us, \3 us per call/' | cut -c1-200 | tee -a $O/inflight.txt
      echo | tee -a $O/inflight.txt
    done
  done
done
