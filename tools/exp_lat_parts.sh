# usage: bash tools/exp_lat_parts.sh "<part bytes> <parts> <prime bytes>" ...  -- experiment: the parse parts of a latency segment
# (HD_LAT_PART_BYTES / HD_LAT_PARTS_MAX / HD_LAT_PRIME_BYTES): parity of the latency paths, one batch of 16 blocks, the hook at 16 callers
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/exp_guard.sh
exp_guard include/hipdeflate_params.h
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
for cfg in "$@"; do
  set -- $cfg
  sed -i "s/#define HD_LAT_PART_BYTES .*/#define HD_LAT_PART_BYTES  $1u/; s/#define HD_LAT_PARTS_MAX .*/#define HD_LAT_PARTS_MAX   $2u/; s/#define HD_LAT_PRIME_BYTES .*/#define HD_LAT_PRIME_BYTES $3u/" include/hipdeflate_params.h
  touch 7bgzf_amd/csrc/hd_api.hip oracle/hd_deflate_twin.c
  make -s -C oracle > /dev/null 2>&1
  make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
  echo "== parts of $1 bytes x $2, primed with $3"
  timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "latency" 2>&1 | tail -1
  for lv in 2 6; do HOOK_LEVEL=$lv HOOK_N=16 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin 0; done
  for m in hip2 hip6; do BGZF_METHOD=$m ./7bgzf_amd/hook_bench /tmp/hook_fq.bin 16 2 | cut -c1-170; done
done
