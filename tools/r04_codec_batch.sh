set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r04_codec; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 300 -p no:cacheprovider > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
: > $O/codec_curve.jsonl
for T in 1 4 16 64; do ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 65280 hip_deflate:1 | tee -a $O/codec_curve.jsonl; done
for T in 16 64; do ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 65280 hip_deflate:6 | tee -a $O/codec_curve.jsonl; HIPDEFLATE_CODEC_BATCH=0 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 65280 hip_deflate:1 | sed 's/hip_deflate:1/hip_deflate:1 (HIPDEFLATE_CODEC_BATCH=0)/' | tee -a $O/codec_curve.jsonl; done
BGZF_METHOD=hip1 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin 16 2 | tee -a $O/codec_curve.jsonl
timeout -k 10 300 bash tools/e2e_cielbox.sh $O/cielbox 512 > $O/cielbox.log 2>&1 || tail -5 $O/cielbox.log
head -6 $O/cielbox/e2e_cielbox.txt
