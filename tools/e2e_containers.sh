# end-to-end (PCIe + stdio inclusive) time of the container hosts on 1 GiB in /dev/shm, next to the
# reference's own applets on the same file and the same host cores
set -e
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
t=s.fastq_like(64<<20)
with open('/dev/shm/hd_in.bin','wb') as f:
    for _ in range(16): f.write(t.tobytes())
PY
IN=/dev/shm/hd_in.bin; E=/dev/shm/err.txt
t() { grep ellapsed $E | sed 's/ellapsed time: //'; }
./7bgzf_amd/hd7dictzip -G1 $IN /dev/shm/o.dz 2>$E; echo "hd7dictzip -G1: $(t)  $(stat -c %s /dev/shm/o.dz) bytes"
./7bgzf_amd/hd7dictzip -d /dev/shm/o.dz 2>$E | cmp - $IN && echo "hd7dictzip -d: $(t) ROUNDTRIP_OK"
./oracle/_ref/cielbox_ref 7dictzip -cl1 -@16 $IN /dev/shm/r.dz 2>$E; echo "reference 7dictzip -l1 -@16: $(t)"
./oracle/_ref/cielbox_ref 7dictzip -cd -@16 /dev/shm/r.dz 2>$E >/dev/null; echo "reference 7dictzip -d -@16: $(t)"
rm -f /dev/shm/o.dz /dev/shm/r.dz
./7bgzf_amd/hd7razf -G1 $IN 2>$E >/dev/shm/o.raz; echo "hd7razf -G1: $(t)  $(stat -c %s /dev/shm/o.raz) bytes"
./7bgzf_amd/hd7razf -d /dev/shm/o.raz 2>$E | cmp - $IN && echo "hd7razf -d: $(t) ROUNDTRIP_OK"
./oracle/_ref/cielbox_ref 7razf -cl1 -@16 $IN 2>$E >/dev/shm/r.raz; echo "reference 7razf -l1 -@16: $(t)"
rm -f /dev/shm/o.raz /dev/shm/r.raz
head -c 209715200 $IN > /dev/shm/hd_in200.bin    # the reference's GZinga reader handles ~2500 blocks
./7bgzf_amd/hd7gzinga -G1 < /dev/shm/hd_in200.bin 2>$E >/dev/shm/o.gz; echo "hd7gzinga -G1 (200 MiB): $(t)  $(stat -c %s /dev/shm/o.gz) bytes"
./7bgzf_amd/hd7gzinga -d /dev/shm/o.gz 2>$E | cmp - /dev/shm/hd_in200.bin && echo "hd7gzinga -d: $(t) ROUNDTRIP_OK"
./oracle/_ref/cielbox_ref 7gzinga -cl1 -@16 < /dev/shm/hd_in200.bin 2>$E >/dev/shm/r.gz; echo "reference 7gzinga -l1 -@16 (200 MiB): $(t)"
rm -f $E $IN /dev/shm/hd_in200.bin /dev/shm/o.gz /dev/shm/r.gz
