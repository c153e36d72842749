# experiment: the reference's thread-per-block loop on the batched hip_deflate with different linger times of the batcher
set -e
cd ${GRAFT_REPO_ROOT:-.}
D=/dev/shm/hd_cb; mkdir -p $D
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
t=s.fastq_like(64<<20, seed=1234).tobytes()
with open('$D/in.bin','wb') as f:
    for _ in range(4): f.write(t)
"
SZ=$(stat -c %s $D/in.bin)
for T in 16 64; do
  t0=$(date +%s.%N); ./oracle/_ref/cielbox_ref 7bgzf -l1 -@$T < $D/in.bin > $D/o.bgz 2>/dev/null; t1=$(date +%s.%N)
  python3 -c "print('ref -l1 -@$T: %.3f GB/s' % ($SZ/($t1-$t0)/1e9))"
  for L in 8 30 80 200; do
    for B in 1 0; do
      t0=$(date +%s.%N); HIPDEFLATE_CODEC_BATCH=$B HIPDEFLATE_LINGER_US=$L ./oracle/_ref/cielbox_hip 7bgzf -G1 -@$T < $D/in.bin > $D/o.bgz 2>/dev/null; t1=$(date +%s.%N)
      python3 -c "print('hip -G1 -@$T batch=$B linger=$L: %.3f GB/s' % ($SZ/($t1-$t0)/1e9))"
    done
  done
done
rm -rf $D
