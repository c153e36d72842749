# usage: bash tools/exp_lat_seg.sh <bytes> ...  -- latency-mode segment size of the dynamic levels: hook throughput / ratio
# at hip2 and hip6 (experiment only: patches include/hipdeflate_params.h, rebuilds, restores)
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/exp_guard.sh
exp_guard include/hipdeflate_params.h
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
s.text_like(64<<20, seed=4321).tofile('/tmp/hook_tx.bin')
"
for seg in "$@"; do
  sed -i "s/#define HD_LAT_SEG_BYTES(level) .*/#define HD_LAT_SEG_BYTES(level) ((level) <= 1 ? 4080u : ${seg}u)/" include/hipdeflate_params.h
  touch 7bgzf_amd/csrc/hd_api.hip
  make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
  echo "== dynamic-level latency segments of $seg bytes"
  for m in hip2 hip6; do
    for f in /tmp/hook_fq.bin /tmp/hook_tx.bin; do
      BGZF_METHOD=$m ./7bgzf_amd/hook_bench $f 16 2 | cut -c1-170
    done
  done
  HOOK_LEVEL=2 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin 0 | head -5 | tail -2
done
