# usage: bash tools/startup.sh <outdir>  -- where the start-up of a small hd7bgzf job goes: the dynamic loader,
# the HIP runtime, the context, the code object, the pipeline's pinned buffers, the job itself, the exit
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=${1:-gpurun_out/startup}; mkdir -p $OUT
D=/dev/shm/hd_su; mkdir -p $D
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
t=s.fastq_like(64<<20, seed=1234).tobytes()
with open('$D/in256.bin','wb') as f:
    for _ in range(4): f.write(t)
"
R=$OUT/startup.txt; : > $R
wall() { local t0=$(date +%s.%N); "${@:2}" 2>> $R; local t1=$(date +%s.%N); python3 -c "print('%-58s wall %.3f s' % ('$1', $t1-$t0))" >> $R; }
export HIPDEFLATE_INIT_TRACE=1 HD7BGZF_TIMING=1
echo "# empty input: start-up + exit alone" >> $R
empty() { ./7bgzf_amd/hd7bgzf -G1 < /dev/null > /dev/null; }
wall "hd7bgzf -G1 < /dev/null (first run on the box)" empty
wall "hd7bgzf -G1 < /dev/null (again)" empty
wall "hd7bgzf -G1 < /dev/null (again)" empty
echo "# the dynamic loader's share (LD_DEBUG=statistics)" >> $R
LD_DEBUG=statistics ./7bgzf_amd/hd7bgzf -G1 < /dev/null 2>&1 > /dev/null | grep -E "total startup time|relocation|load" | head -4 >> $R
echo "# 256 MiB file to file" >> $R
enc() { ./7bgzf_amd/hd7bgzf -G1 -@8 -i $D/in256.bin -o $D/out256.bgz; }
dec() { ./7bgzf_amd/hd7bgzf -d -@8 -i $D/out256.bgz -o $D/back256.bin; }
wall "hd7bgzf -G1 -@8 256 MiB file to file" enc
wall "hd7bgzf -G1 -@8 256 MiB file to file (again)" enc
wall "hd7bgzf -d -@8 256 MiB file to file" dec
wall "hd7bgzf -d -@8 256 MiB file to file (again)" dec
cmp $D/back256.bin $D/in256.bin && echo "decode == input" >> $R
if [ -x oracle/_ref/cielbox_ref ]; then
  refenc() { ./oracle/_ref/cielbox_ref 7bgzf -l1 -@16 < $D/in256.bin > $D/ref.bgz; }
  wall "cielbox_ref 7bgzf -l1 -@16 256 MiB (the CPU reference)" refenc
fi
rm -rf $D
cat $R
