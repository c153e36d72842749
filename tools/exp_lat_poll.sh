# usage: bash tools/exp_lat_poll.sh  -- A/B on one box: hipdeflate_lat_run waiting in hipStreamSynchronize or polling the word the run's
# last kernel stores (HIPDEFLATE_LAT_POLL=1); one batch alone and the hook at 8 / 16 / 64 callers
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
for P in 0 1 0 1; do
  export HIPDEFLATE_LAT_POLL=$P
  echo "== HIPDEFLATE_LAT_POLL=$P"
  for lv in 1 2; do HOOK_LEVEL=$lv HOOK_N=16 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin 0; done
  for m in hip1 hip2 hip6; do for T in 8 16; do BGZF_METHOD=$m ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 | cut -c1-170; done; done
done
