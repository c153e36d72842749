# usage: bash tools/exp_seg_prime.sh  -- A/B on one box: latency segments primed with the end of their predecessor (HD_LAT_SEG_PRIME) or not
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
for P in 1 0 1 0; do
  touch 7bgzf_amd/csrc/hd_api.hip
  make -s -C 7bgzf_amd/csrc EXTRA=-DHD_LAT_SEG_PRIME=$P > /dev/null 2>&1
  echo "== HD_LAT_SEG_PRIME=$P"
  for lv in 1 2; do HOOK_LEVEL=$lv HOOK_N=16 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin 0; done
  for m in hip1 hip2; do BGZF_METHOD=$m ./7bgzf_amd/hook_bench /tmp/hook_fq.bin 16 2 | cut -c1-170; done
done
touch 7bgzf_amd/csrc/hd_api.hip
make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
