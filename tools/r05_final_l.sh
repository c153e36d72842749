# round 5: the reference's CLI on the batch pipelines (7bgzf and 7migz, both ways) -- the tests
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_l; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_cielbox_hip.py -q -m gpu -x --timeout 700 > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
D=/dev/shm/hd_mz; mkdir -p $D
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
t=s.text_like(64<<20, seed=77).tobytes()
with open('$D/in.bin','wb') as f:
    for _ in range(8): f.write(t)
"
: > $O/e2e_migz.txt
tm() { local t0=$(date +%s.%N); "${@:2}" 2>> $O/e2e_migz_stderr.txt; local t1=$(date +%s.%N); python3 -c "print('$1: %.3f s  %.3f GB/s' % ($t1-$t0, 536870912/($t1-$t0)/1e9))" >> $O/e2e_migz.txt; }
e_hip() { ./oracle/_ref/cielbox_hip 7migz -G6 -b1024 -@16 < $D/in.bin > $D/hip.mgz; }
e_hip_pb() { HIP_DEFLATE_PER_BLOCK=1 ./oracle/_ref/cielbox_hip 7migz -G6 -b1024 -@16 < $D/in.bin > $D/hip_pb.mgz; }
e_ref() { ./oracle/_ref/cielbox_ref 7migz -l6 -b1024 -@16 < $D/in.bin > $D/ref.mgz; }
d_hip() { ./oracle/_ref/cielbox_hip 7migz -d -@16 < $D/ref.mgz > $D/back_hip.bin; }
d_hip_pb() { HIP_INFLATE_PER_BLOCK=1 ./oracle/_ref/cielbox_hip 7migz -d -@16 < $D/ref.mgz > $D/back_hip_pb.bin; }
d_ref() { ./oracle/_ref/cielbox_ref 7migz -d -@16 < $D/ref.mgz > $D/back_ref.bin; }
tm "cielbox_hip 7migz -G6 -b1024 -@16 (batched), 512 MiB of text" e_hip
tm "cielbox_hip 7migz -G6 -b1024 -@16, HIP_DEFLATE_PER_BLOCK=1" e_hip_pb
tm "cielbox_ref 7migz -l6 -b1024 -@16 (libdeflate 6)" e_ref
tm "cielbox_hip 7migz -d -@16 (batched, the file of the reference)" d_hip
tm "cielbox_hip 7migz -d -@16, HIP_INFLATE_PER_BLOCK=1" d_hip_pb
tm "cielbox_ref 7migz -d -@16" d_ref
cmp $D/back_hip.bin $D/in.bin && cmp $D/back_hip_pb.bin $D/in.bin && cmp $D/hip.mgz $D/hip_pb.mgz && echo "decodes == input; the batched file == the per-block file" >> $O/e2e_migz.txt
ls -l $D/*.mgz | awk '{print $5, $9}' >> $O/e2e_migz.txt
rm -rf $D
cat $O/e2e_migz.txt
