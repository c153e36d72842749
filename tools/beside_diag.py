"""diagnostic for the emit-beside-parse scheme: one launch of N blocks at level 6 through the host API, its wall time, the stall counter,
kernel == twin.  usage: python tools/beside_diag.py [N=600] [block_bytes=65280]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hdtest
pkg = importlib.import_module("7bgzf_amd")
synth = hdtest.synth()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
bb = int(sys.argv[2]) if len(sys.argv) > 2 else 65280
data = bytes(synth.fastq_like(n * bb, seed=3))
offs = [i * bb for i in range(n)]
lens = [bb] * n
slot = int(pkg.lib().hipdeflate_bound(bb, 6))
for rep in range(3):
    s0 = int(pkg.lib().hipdeflate_stall_count())
    t0 = time.time()
    members, crc, st = pkg.batch_deflate(data, offs, lens, 6, pkg.FRAME_RAW, slot=slot)
    dt = time.time() - t0
    stalls = int(pkg.lib().hipdeflate_stall_count()) - s0
    print("rep %d: %d blocks in %.3f s, stalls %d, statuses nonzero %d" % (rep, n, dt, stalls, sum(1 for x in st if x)), flush=True)
bad = 0
for i in range(0, n, max(1, n // 40)):
    r, t = hdtest.codec_twin(data[offs[i]:offs[i] + bb], 6, cap=slot)
    bad += not (r == 0 and t == members[i])
print("twin comparison on %d sampled blocks: %d bad" % (len(range(0, n, max(1, n // 40))), bad))
