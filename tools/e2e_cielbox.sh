# usage: bash tools/e2e_cielbox.sh <outdir> [MiB=512]  -- the REFERENCE'S OWN CLI on the hip backend (oracle/_ref/cielbox_hip: the
# reference built with integration/7bgzf-hip.patch) beside the unpatched reference (cielbox_ref), same file, tmpfs to tmpfs:
# `7bgzf -@16` encode at level 1 / 6 and `7bgzf -d -@16` -- the thread-per-block loops of applet/7bgzf.c driving one block per call
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=$1; MIB=${2:-512}
mkdir -p $OUT
D=/dev/shm/hd_cb; mkdir -p $D
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
t=s.fastq_like(64<<20, seed=1234).tobytes()
with open('$D/in.bin','wb') as f:
    for _ in range($MIB//64): f.write(t)
"
SZ=$(stat -c %s $D/in.bin)
: > $OUT/e2e_cielbox.txt
tm() { local t0=$(date +%s.%N); "${@:2}" 2>> $OUT/e2e_cielbox_stderr.txt; local t1=$(date +%s.%N); python3 -c "print('$1: %.3f s  %.3f GB/s' % ($t1-$t0, $SZ/($t1-$t0)/1e9))" >> $OUT/e2e_cielbox.txt; }
enc_hip1() { ./oracle/_ref/cielbox_hip 7bgzf -G1 -@16 < $D/in.bin > $D/hip1.bgz; }
enc_hip6() { ./oracle/_ref/cielbox_hip 7bgzf -G6 -@16 < $D/in.bin > $D/hip6.bgz; }
enc_hip6_pb() { HIP_DEFLATE_PER_BLOCK=1 ./oracle/_ref/cielbox_hip 7bgzf -G6 -@16 < $D/in.bin > $D/hip6_pb.bgz; }
enc_hip1_pb() { HIP_DEFLATE_PER_BLOCK=1 ./oracle/_ref/cielbox_hip 7bgzf -G1 -@16 < $D/in.bin > $D/hip1_pb.bgz; }
enc_ref1() { ./oracle/_ref/cielbox_ref 7bgzf -l1 -@16 < $D/in.bin > $D/ref1.bgz; }
enc_ref6() { ./oracle/_ref/cielbox_ref 7bgzf -l6 -@16 < $D/in.bin > $D/ref6.bgz; }
dec_hip() { ./oracle/_ref/cielbox_hip 7bgzf -d -@16 < $D/ref6.bgz > $D/back_hip.bin; }
dec_hip_pb() { HIP_INFLATE_PER_BLOCK=1 ./oracle/_ref/cielbox_hip 7bgzf -d -@16 < $D/ref6.bgz > $D/back_hip_pb.bin; }
dec_ref() { ./oracle/_ref/cielbox_ref 7bgzf -d -@16 < $D/ref6.bgz > $D/back_ref.bin; }
tm "cielbox_hip 7bgzf -G1 -@16 (the batched loop of the patch on hipdeflate_pipe)" enc_hip1
tm "cielbox_hip 7bgzf -G1 -@16, HIP_DEFLATE_PER_BLOCK=1 (the loop of the reference, hip_deflate per block)" enc_hip1_pb
tm "cielbox_ref 7bgzf -l1 -@16 (libdeflate 1)" enc_ref1
tm "cielbox_hip 7bgzf -G6 -@16 (batched)" enc_hip6
tm "cielbox_hip 7bgzf -G6 -@16, HIP_DEFLATE_PER_BLOCK=1" enc_hip6_pb
tm "cielbox_ref 7bgzf -l6 -@16 (libdeflate 6)" enc_ref6
tm "cielbox_hip 7bgzf -d -@16 (the batched loop of the patch on hipdeflate_unpipe, libdeflate-6 file)" dec_hip
tm "cielbox_hip 7bgzf -d -@16 again" dec_hip
tm "cielbox_hip 7bgzf -d -@16, HIP_INFLATE_PER_BLOCK=1 (the loop of the reference, hip_inflate per member)" dec_hip_pb
tm "cielbox_ref 7bgzf -d -@16 (its own inflater)" dec_ref
cmp $D/back_hip.bin $D/in.bin && cmp $D/back_hip_pb.bin $D/in.bin && cmp $D/back_ref.bin $D/in.bin && echo "all three decodes == input" >> $OUT/e2e_cielbox.txt
cmp $D/hip6.bgz $D/hip6_pb.bgz && echo "-G6: the batched file == the per-block file" >> $OUT/e2e_cielbox.txt
ls -l $D/*.bgz | awk '{print $5, $9}' >> $OUT/e2e_cielbox.txt
rm -rf $D
cat $OUT/e2e_cielbox.txt
