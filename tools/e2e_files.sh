# usage: bash tools/e2e_files.sh <outdir> [GiB=4]  -- end-to-end file-to-file rate of hd7bgzf from and to tmpfs (page-cache speed),
# beside the stdin/stdout filter and the reference's 7bgzf -@16 on the same file
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=$1; GIB=${2:-4}
mkdir -p $OUT
D=/dev/shm/hd_e2e; mkdir -p $D
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
t=s.fastq_like(256<<20, seed=1234).tobytes()
with open('$D/in.bin','wb') as f:
    for _ in range($GIB*4): f.write(t)
"
ls -la $D/in.bin
: > $OUT/e2e_files.txt
tm() { local t0=$(date +%s.%N); "${@:2}"; local t1=$(date +%s.%N); echo "$1: $(python3 -c "print('%.3f s  %.2f GB/s in' % ($t1-$t0, $3/($t1-$t0)/1e9))")" >> $OUT/e2e_files.txt; }
SZ=$(stat -c %s $D/in.bin)
# (the output is removed before every timed run: truncating the previous run's gigabytes in tmpfs was a quarter of a second of the next)
run_files() { rm -f $D/out.bgz; HD7BGZF_TIMING=1 ./7bgzf_amd/hd7bgzf -G$1 -@$2 -i $D/in.bin -o $D/out.bgz 2>> $OUT/e2e_stderr.txt; }
run_filter() { ./7bgzf_amd/hd7bgzf -G1 < $D/in.bin > $D/out2.bgz 2>> $OUT/e2e_stderr.txt; }
run_ref() { ./oracle/_ref/cielbox_ref 7bgzf -l1 -@16 < $D/in1g.bin > $D/ref.bgz 2>> $OUT/e2e_stderr.txt; }
for T in 4 8 16; do
  for L in 1 6; do
    t0=$(date +%s.%N); run_files $L $T; t1=$(date +%s.%N)
    python3 -c "print('hd7bgzf -G$L -@$T file-to-file: %.3f s  %.2f GB/s in' % ($t1-$t0, $SZ/($t1-$t0)/1e9))" >> $OUT/e2e_files.txt
  done
done
# decode, file to file (the level-1 file of the last run above is level 6's: make the level-1 one again)
run_files 1 8
for T in 4 16; do
  rm -f $D/back.bin
  t0=$(date +%s.%N); ./7bgzf_amd/hd7bgzf -d -@$T -i $D/out.bgz -o $D/back.bin 2>> $OUT/e2e_stderr.txt; t1=$(date +%s.%N)
  python3 -c "print('hd7bgzf -d -@$T file-to-file: %.3f s  %.2f GB/s out' % ($t1-$t0, $SZ/($t1-$t0)/1e9))" >> $OUT/e2e_files.txt
done
cmp $D/back.bin $D/in.bin && echo "decode output == input" >> $OUT/e2e_files.txt
rm -f $D/back.bin
# a 256 MiB file: what the start-up costs (context, pinning)
head -c $((256<<20)) $D/in.bin > $D/in256.bin
for L in 1 6; do
  t0=$(date +%s.%N); ./7bgzf_amd/hd7bgzf -G$L -@8 -i $D/in256.bin -o $D/out256.bgz 2>> $OUT/e2e_stderr.txt; t1=$(date +%s.%N)
  python3 -c "print('hd7bgzf -G$L -@8 256 MiB file-to-file: %.3f s  %.2f GB/s in' % ($t1-$t0, (256<<20)/($t1-$t0)/1e9))" >> $OUT/e2e_files.txt
done
t0=$(date +%s.%N); ./7bgzf_amd/hd7bgzf -d -@8 -i $D/out256.bgz -o $D/back256.bin 2>> $OUT/e2e_stderr.txt; t1=$(date +%s.%N)
python3 -c "print('hd7bgzf -d -@8 256 MiB file-to-file: %.3f s  %.2f GB/s out' % ($t1-$t0, (256<<20)/($t1-$t0)/1e9))" >> $OUT/e2e_files.txt
cmp $D/back256.bin $D/in256.bin && echo "256 MiB decode output == input" >> $OUT/e2e_files.txt
rm -f $D/in256.bin $D/out256.bgz $D/back256.bin
t0=$(date +%s.%N); run_filter; t1=$(date +%s.%N)
python3 -c "print('hd7bgzf -G1 filter (stdin/stdout): %.3f s  %.2f GB/s in' % ($t1-$t0, $SZ/($t1-$t0)/1e9))" >> $OUT/e2e_files.txt
run_files 1 8; cmp $D/out.bgz $D/out2.bgz && echo "file-to-file output == filter output" >> $OUT/e2e_files.txt
if [ -x oracle/_ref/cielbox_ref ]; then
  head -c $((1<<30)) $D/in.bin > $D/in1g.bin
  t0=$(date +%s.%N); run_ref; t1=$(date +%s.%N)
  python3 -c "print('reference 7bgzf -l1 -@16 (1 GiB): %.3f s  %.2f GB/s in' % ($t1-$t0, (1<<30)/($t1-$t0)/1e9))" >> $OUT/e2e_files.txt
fi
rm -rf $D
cat $OUT/e2e_files.txt
