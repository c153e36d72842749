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
for T in 4 8 16; do
  for L in 1 6; do
    /usr/bin/time -f "hd7bgzf -G$L -@$T file-to-file: %e s" ./7bgzf_amd/hd7bgzf -G$L -@$T -i $D/in.bin -o $D/out.bgz 2>> $OUT/e2e_files.txt
    ls -la $D/out.bgz >> $OUT/e2e_files.txt
  done
done
/usr/bin/time -f "hd7bgzf -G1 filter (stdin/stdout): %e s" ./7bgzf_amd/hd7bgzf -G1 < $D/in.bin > $D/out2.bgz 2>> $OUT/e2e_files.txt
cmp $D/out.bgz $D/out2.bgz || true
if [ -x oracle/_ref/cielbox_ref ]; then
  head -c $((1<<30)) $D/in.bin > $D/in1g.bin
  /usr/bin/time -f "reference 7bgzf -l1 -@16 (1 GiB): %e s" ./oracle/_ref/cielbox_ref 7bgzf -l1 -@16 < $D/in1g.bin > $D/ref.bgz 2>> $OUT/e2e_files.txt
fi
rm -rf $D
grep -E " s$|ellapsed" $OUT/e2e_files.txt
