"""One-off differential run on the GPU box for the workgroup levels (3..9): block lengths around everything the kernel counts in --
the six-byte key (0..8), a step (62..66, 126..130), a piece (1018..1030, 2046..2050), the verified span near a piece's end, sixteen
steps short of a piece, a 0xff00-byte block, the 64 KiB ring (65530..65542, 131070..131074) and its second lap, 1 MiB members -- of four
kinds of data (text, FASTQ-like, one byte repeated, noise): kernel bytes == twin bytes, zlib inflates them.
usage: python tools/big_fuzz_wg.py [seed]"""
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
s = hdtest.synth()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 77
rng = np.random.default_rng(seed)
sizes = list(range(0, 9)) + list(range(62, 67)) + list(range(126, 131)) + list(range(1007, 1031)) + list(range(2046, 2051)) + \
    [4095, 4096, 4097, 15 * 1024 + 1, 16 * 1024, 24 * 1024 + 5, 25 * 1024, 33 * 1024 - 1, 65279, 65280, 65281] + \
    list(range(65530, 65543)) + [98304, 131070, 131072, 131074, 200001, 262144 + 7, 1 << 20, (1 << 20) + 1023]
big = {"text": bytes(s.text_like(1 << 21, seed=seed)), "fastq": bytes(s.fastq_like(1 << 21, seed=seed + 1)),
       "run": bytes([65]) * (1 << 21), "noise": bytes(s.random_bytes(1 << 21, seed=seed + 2)),
       "mix": bytes(s.text_like(300000, seed=seed + 3)) + bytes(700) + bytes(s.random_bytes(5000, seed=seed + 4)) * 3 +
              bytes(s.fastq_like(1 << 20, seed=seed + 5)) + bytes([7, 8]) * 400000}
blocks, names = [], []
for kind, data in big.items():
    for n in sizes:
        o = int(rng.integers(0, len(data) - n + 1)) if n <= len(data) else 0
        blocks.append(data[o:o + n])
        names.append("%s/%d" % (kind, n))
blob, offs, lens = bytearray(), [], []
for b in blocks:
    offs.append(len(blob))
    lens.append(len(b))
    blob += b + bytes(-len(b) % 16)
blob = bytes(blob)
t0 = time.time()
total = bad = 0
for level in (3, 4, 5, 6, 9):
    for frame, twin_fn in ((pkg.FRAME_RAW, hdtest.oracle_twin), (pkg.FRAME_RAW_FLUSH, hdtest.oracle_twin_flush)):
        slot = int(pkg.lib().hipdeflate_bound(max(lens), level))
        members, crc, st = pkg.batch_deflate(blob, offs, lens, level, frame, slot=slot)
        with ThreadPoolExecutor(min(32, os.cpu_count() or 16)) as ex:
            twins = list(ex.map(lambda b: twin_fn(b, level, cap=slot), blocks))
        for i, b in enumerate(blocks):
            total += 1
            ok = st[i] == 0 and twins[i][0] == 0 and members[i] == twins[i][1] and int(crc[i]) == zlib.crc32(b)
            if ok:
                tail = b"\x03\x00" if frame == pkg.FRAME_RAW_FLUSH else b""
                ok = zlib.decompressobj(-15).decompress(members[i] + tail) == b
            if not ok:
                bad += 1
                print("MISMATCH %s level %d frame %d st %d" % (names[i], level, frame, st[i]), flush=True)
        print("level %d frame %d done, %d so far, %d bad, %.0f s" % (level, frame, total, bad, time.time() - t0), flush=True)
stalls = int(pkg.lib().hipdeflate_stall_count())
print("BIG_FUZZ_WG %s: %d comparisons, %d bad, %d stalls" % ("OK" if bad == 0 and stalls == 0 else "FAILED", total, bad, stalls))
sys.exit(1 if bad or stalls else 0)
