# usage: bash tools/exp_hook_inflight.sh  -- experiment: hook throughput against the number of batches allowed on the device at once
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
export HIPDEFLATE_HOOK_STATS=1
for M in hip2 hip1; do
  for T in 8 16 32 64; do
    for F in 1 2 3 8; do
      echo -n "$M T=$T inflight=$F: "
      BGZF_METHOD=$M HIPDEFLATE_INFLIGHT=$F ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 1.5 2>&1 | tr '\n' ' ' | sed 's/.*batches (\([0-9.]*\) blocks each).*device \([0-9.]*\).*"GBps_in": \([0-9.]*\).*/blocks per batch \1, device \2 us, \3 GB\/s/'
      echo
    done
  done
done
