# usage: bash tools/hook_curve.sh <outdir> [quick]  -- hook throughput at 1/4/8/16/64 caller threads, ours (hip1, hip6) and the reference's
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=$1
mkdir -p $OUT
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
: > $OUT/hook_curve.jsonl
./7bgzf_amd/hook_bench /tmp/hook_fq.bin 0 >> $OUT/hook_curve.jsonl
HOOK_LEVEL=6 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin 0 >> $OUT/hook_curve.jsonl
for T in 1 4 8 16 32 64; do
  HIPDEFLATE_HOOK_STATS=1 BGZF_METHOD=hip1 ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 >> $OUT/hook_curve.jsonl 2>> $OUT/hook_stats.txt
done
for T in 8 16; do
  for M in hip2 hip5 hip6; do
    HIPDEFLATE_HOOK_STATS=1 BGZF_METHOD=$M ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 >> $OUT/hook_curve.jsonl 2>> $OUT/hook_stats.txt
  done
done
if [ -x oracle/_ref/hook_bench_ref ] && [ "$2" != quick ]; then
  for T in 1 4 8 16 32 64; do
    BGZF_METHOD=libdeflate1 ./oracle/_ref/hook_bench_ref /tmp/hook_fq.bin $T 2 >> $OUT/hook_curve.jsonl
  done
  BGZF_METHOD=libdeflate6 ./oracle/_ref/hook_bench_ref /tmp/hook_fq.bin 16 2 >> $OUT/hook_curve.jsonl
  # the reference's DEFAULT (BGZF_METHOD unset = its zlib at level 6, bgzf_compress.c:54,102): what an unset variable replaces;
  # beside it ours with the variable unset (hip level 6)
  for T in 8 16 64; do
    env -u BGZF_METHOD ./oracle/_ref/hook_bench_ref /tmp/hook_fq.bin $T 2 | sed 's/"method": ""/"method": "(unset: reference zlib6)"/' >> $OUT/hook_curve.jsonl
    env -u BGZF_METHOD ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 | sed 's/"method": ""/"method": "(unset: hip6)"/' >> $OUT/hook_curve.jsonl
  done
fi
cat $OUT/hook_curve.jsonl
cat $OUT/hook_stats.txt
