"""One-off wide differential run of the inflate kernel on the GPU box: mutated streams (bit flips, byte
sets, cuts, spliced tails) of zlib's and our own encoders' output over tests/hdtest.corpus_fuzz blocks;
the kernel's verdict code, bytes and CRC-32 must equal the oracle's (oracle/hd_inflate.c, itself pinned
on libdeflate_inflate), in the strict and in the flushed-chunk mode.
HD_FUZZ_PER_CALL=1: the same streams through hip_inflate / hip_inflate_flush from 32 threads instead -- the per-call boundary, i.e.
the two-wavefront latency kernel (hd_inflate_lat.hpp), whose verdicts and bytes must be the batch kernel's, i.e. the oracle's.
usage: python tools/big_fuzz_inflate.py [mutants_per_base] [seeds...]"""
import importlib
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hdtest  # noqa: E402
import numpy as np  # noqa: E402

pkg = importlib.import_module("7bgzf_amd")
per = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seeds = [int(a) for a in sys.argv[2:]] or [21, 22]
t0 = time.time()
total = bad = accepted = 0
for seed in seeds:
    rng = np.random.default_rng(seed)
    blocks = [b for b in hdtest.corpus_fuzz(seed, 140) if 0 < len(b) <= 20000][:60]
    bases = []
    for i, b in enumerate(blocks):
        lvl, strat = [(1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)][i % 6]
        c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 9, strat)
        bases.append((c.compress(b) + c.flush(), len(b), False))
        r, z = hdtest.oracle_twin(b, [1, 2, 6][i % 3])
        bases.append((z, len(b), False))
        r, z = hdtest.oracle_twin_flush(b, [1, 6][i % 2])
        bases.append((z, len(b), True))
    for flushed in (False, True):
        streams, caps = [], []
        for z, n, is_flush in bases:
            if is_flush != flushed and not flushed:
                continue                    # strict mode: final-block streams only (a chunk is refused outright)
            for _ in range(per):
                m = bytearray(z)
                kind = int(rng.integers(0, 6))
                if kind < 3 and m:
                    for _ in range(kind + 1):
                        bit = int(rng.integers(0, len(m) * 8))
                        m[bit >> 3] ^= 1 << (bit & 7)
                elif kind == 3 and len(m) > 1:
                    m = m[: int(rng.integers(1, len(m)))]
                elif kind == 4 and m:
                    m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
                else:
                    m += bytes(rng.integers(0, 256, int(rng.integers(0, 12)), dtype=np.uint8))
                streams.append(bytes(m))
                caps.append(max(0, n + int(rng.integers(-2, 3)) * 40))
        if os.environ.get("HD_FUZZ_PER_CALL") == "1":
            call = pkg.hip_inflate_flush if flushed else pkg.hip_inflate
            with ThreadPoolExecutor(32) as ex:
                got = list(ex.map(lambda a: call(a[0], a[1]), zip(streams, caps)))
            st = [g[0] for g in got]
            outs = [g[1] for g in got]
            crc = [zlib.crc32(o) for o in outs]                # (the per-call decoder hands no CRC back)
        else:
            outs, crc, st = pkg.batch_inflate(streams, caps, flushed=flushed)
        fn = hdtest.oracle_inflate_flushed if flushed else hdtest.oracle_inflate
        with ThreadPoolExecutor(min(64, os.cpu_count() or 16)) as ex:
            want = list(ex.map(lambda a: fn(a[0], a[1]), zip(streams, caps)))
        for i in range(len(streams)):
            total += 1
            r, o = want[i]
            ok = int(st[i]) == r and (r != 0 or (outs[i] == o and int(crc[i]) == zlib.crc32(o)))
            accepted += r == 0
            if not ok:
                bad += 1
                print("MISMATCH seed %d flushed %d #%d: kernel %d oracle %d" % (seed, flushed, i, int(st[i]), r), flush=True)
        print("seed %d flushed=%d done: %d streams so far, %d accepted, %d bad, %.0f s" % (seed, flushed, total, accepted, bad, time.time() - t0), flush=True)
print("BIG_FUZZ_INFLATE %s: %d streams, %d accepted by both, %d bad" % ("OK" if bad == 0 else "FAILED", total, accepted, bad))
sys.exit(1 if bad else 0)
