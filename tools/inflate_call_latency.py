"""what one hip_inflate call (the zlibutil_code_dec boundary, lib/zlibutil.h:47) on one 0xff00-byte block costs: a DEFLATE stream is one
wavefront's serial work however empty the chip is.  usage: python tools/inflate_call_latency.py"""
import importlib, sys, time, zlib
sys.path.insert(0, '.')
pkg = importlib.import_module("7bgzf_amd")
synth = importlib.import_module("7bgzf_amd.synth")
blk = bytes(synth.fastq_like(0xff00, seed=5))
for name, z in (("zlib-6 stream", zlib.compress(blk, 6)[2:-4]), ("own level-1 stream", pkg.hip_deflate(blk, 1)[1])):
    r, out = pkg.hip_inflate(z, 0xff00)
    assert r == 0 and out == blk
    t0 = time.time()
    for _ in range(200):
        pkg.hip_inflate(z, 0xff00)
    us = (time.time() - t0) / 200 * 1e6
    print("hip_inflate, one 0xff00-byte block per call, %s: %.0f us per call (%.3f GB/s out per caller)" % (name, us, 0xff00 / us / 1e3))
