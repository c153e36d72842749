"""One-off wide differential run on the GPU box (not part of the test suite: minutes, not seconds):
for many seeds of tests/hdtest.corpus_fuzz (+ corpus_phrases) and every level class, kernel bytes == twin bytes, zlib
inflates them, and the kernel's own inflate returns the input; long blocks (flushed segments) too.
usage: python tools/big_fuzz.py [blocks_per_seed] [seeds...]"""
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
count = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seeds = [int(a) for a in sys.argv[2:]] or [11, 12, 13, 14]
t0 = time.time()
total = bad = 0
for seed in seeds:
    blocks = hdtest.corpus_fuzz(seed, count) + hdtest.corpus_phrases(seed, count // 8)
    rng = np.random.default_rng(seed)
    # a few long blocks: random length 330..900 KB made of fuzz blocks back to back
    for _ in range(3):
        n = int(rng.integers(330_000, 900_000))
        cat = b"".join(blocks[int(i)] for i in rng.integers(0, len(blocks), 40))
        blocks.append((cat * (n // max(len(cat), 1) + 1))[:n])
    blob, offs, lens = bytearray(), [], []
    for b in blocks:
        offs.append(len(blob))
        lens.append(len(b))
        blob += b + bytes(-len(b) % 16)
    blob = bytes(blob)
    for level in (1, 2, 3, 4, 5, 6, 7, 8, 9):
        for frame, twin_fn in ((pkg.FRAME_RAW, hdtest.oracle_twin), (pkg.FRAME_RAW_FLUSH, hdtest.oracle_twin_flush),
                               (pkg.FRAME_RAW | pkg.FRAME_LATENCY, hdtest.codec_twin),
                               (pkg.FRAME_RAW_FLUSH | pkg.FRAME_LATENCY, hdtest.codec_twin_flush)):
            slot = int(pkg.lib().hipdeflate_bound(max(lens), level))
            members, crc, st = pkg.batch_deflate(blob, offs, lens, level, frame, slot=slot)
            with ThreadPoolExecutor(min(64, os.cpu_count() or 16)) as ex:
                twins = list(ex.map(lambda b: twin_fn(b, level, cap=slot), blocks))
            for i, b in enumerate(blocks):
                total += 1
                ok = st[i] == 0 and twins[i][0] == 0 and members[i] == twins[i][1] and int(crc[i]) == zlib.crc32(b)
                if ok:
                    tail = b"\x03\x00" if (frame & 0xff) == pkg.FRAME_RAW_FLUSH else b""
                    ok = zlib.decompressobj(-15).decompress(members[i] + tail) == b
                if not ok:
                    bad += 1
                    print("MISMATCH seed %d block %d len %d level %d frame %d st %d" % (seed, i, len(b), level, frame, st[i]), flush=True)
            if (frame & 0xff) == pkg.FRAME_RAW:
                outs, dcrc, dst = pkg.batch_inflate(members, lens)
                for i, b in enumerate(blocks):
                    if dst[i] != 0 or outs[i] != b:
                        bad += 1
                        print("INFLATE MISMATCH seed %d block %d level %d" % (seed, i, level), flush=True)
        print("seed %d level %d done, %d blocks so far, %d bad, %.0f s" % (seed, level, total, bad, time.time() - t0), flush=True)
print("BIG_FUZZ %s: %d comparisons, %d bad" % ("OK" if bad == 0 else "FAILED", total, bad))
sys.exit(1 if bad else 0)
