# end-to-end (PCIe + stdio inclusive) rate of the hd7bgzf host filter on 2 GiB in /dev/shm
set -e
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
t=s.fastq_like(64<<20)
with open('/dev/shm/hd_in.bin','wb') as f:
    for _ in range(32): f.write(t.tobytes())
PY
ls -la /dev/shm/hd_in.bin
for lvl in 1 3; do
  ./7bgzf_amd/hd7bgzf -G$lvl < /dev/shm/hd_in.bin 2>/dev/shm/err.txt > /dev/shm/hd_out.bgz; echo "level $lvl encode: $(grep ellapsed /dev/shm/err.txt)"
  ls -la /dev/shm/hd_out.bgz
done
./7bgzf_amd/hd7bgzf -d < /dev/shm/hd_out.bgz 2>/dev/shm/err.txt > /dev/shm/hd_back.bin; echo "decode: $(grep ellapsed /dev/shm/err.txt)"
cmp /dev/shm/hd_in.bin /dev/shm/hd_back.bin && echo ROUNDTRIP_OK
./oracle/_ref/cielbox_ref 7bgzf -l1 -@16 < /dev/shm/hd_in.bin 2>/dev/shm/err.txt > /dev/shm/ref_out.bgz; echo "reference cielbox 7bgzf -l1 -@16: $(grep ellapsed /dev/shm/err.txt)"
rm -f /dev/shm/err.txt /dev/shm/hd_in.bin /dev/shm/hd_out.bgz /dev/shm/hd_back.bin /dev/shm/ref_out.bgz
