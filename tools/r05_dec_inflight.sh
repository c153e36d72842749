# round 5: hip_inflate's batches in flight (HIPDEFLATE_INFLATE_INFLIGHT) under the reference's -d -@16 loop and under persistent callers
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_dec_inflight; mkdir -p $O; : > $O/ab.txt
D=/dev/shm/hd_cb3; mkdir -p $D
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
t=s.fastq_like(64<<20, seed=1234).tobytes()
with open('$D/in.bin','wb') as f:
    for _ in range(8): f.write(t)
"
./oracle/_ref/cielbox_ref 7bgzf -l6 -@16 < $D/in.bin > $D/ref6.bgz 2>/dev/null
for F in 2 3 4 8 2; do
  for LG in 60 150; do
    t0=$(date +%s.%N); HIPDEFLATE_INFLATE_INFLIGHT=$F HIPDEFLATE_INFLATE_LINGER_US=$LG ./oracle/_ref/cielbox_hip 7bgzf -d -@16 < $D/ref6.bgz > $D/back.bin 2>/dev/null; t1=$(date +%s.%N)
    python3 -c "print('inflight $F linger $LG: cielbox_hip 7bgzf -d -@16 512 MiB %.3f s  %.3f GB/s' % ($t1-$t0, (512<<20)/($t1-$t0)/1e9))" | tee -a $O/ab.txt
  done
done
cmp $D/back.bin $D/in.bin && echo "decode == input" | tee -a $O/ab.txt
rm -rf $D
for F in 2 4; do
  echo "== persistent callers, inflight $F" | tee -a $O/ab.txt
  HIPDEFLATE_INFLATE_INFLIGHT=$F timeout -k 10 300 python3 tools/inflate_call_latency.py $O/icl_$F.jsonl | grep zlib6 | grep -v DEVICES | cut -c1-120 | tee -a $O/ab.txt
done
