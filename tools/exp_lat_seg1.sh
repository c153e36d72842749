# usage: bash tools/exp_lat_seg1.sh <bytes> ...  -- experiment, same box: the level-1 latency segment (HD_LAT_SEG_BYTES(1); a BGZF member
# has room for 22 segments' worst case at most: 2976 bytes and up): parity of the latency paths, one batch of 16 blocks, the hook at 8 / 16 / 64 callers
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
for seg in "$@"; do
  sed -i "s/#define HD_LAT_SEG_BYTES(level) ((level) <= 1 ? [0-9]*u : 8160u)/#define HD_LAT_SEG_BYTES(level) ((level) <= 1 ? ${seg}u : 8160u)/" include/hipdeflate_params.h
  touch 7bgzf_amd/csrc/hd_api.hip oracle/hd_deflate_twin.c
  make -s -C oracle > /dev/null 2>&1
  make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
  echo "== level-1 latency segments of $seg bytes"
  timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "latency or hook or codec" 2>&1 | tail -1
  HOOK_LEVEL=1 HOOK_N=16 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin 0
  for T in 8 16 64; do BGZF_METHOD=hip1 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 | cut -c1-170; done
done
