"""Wide differential run of LATENCY mode on the GPU box (round 3: parse parts, primed parts and segments): block lengths
around every border the mode knows (segment 4080 / 8160, part 2048, prime 512, the 64-byte priming threshold) and random
ones, FASTQ-like / text / noise / fuzz content, levels 1..9, plain and flush form: kernel bytes == twin bytes, zlib inflates
them, CRC-32 right.  usage: python tools/big_fuzz_lat.py [random_blocks_per_seed] [seeds...]"""
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
synth = hdtest.synth()
count = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seeds = [int(a) for a in sys.argv[2:]] or [31, 32, 33]
edges = sorted(set(b + d for b in (512, 2048, 4080, 4096, 6144, 8160, 8192, 12240, 16320, 24480, 32640, 0xff00)
                   for d in (-65, -64, -63, -17, -16, -15, -1, 0, 1, 15, 16, 17, 63, 64, 65, 511, 512, 513) if b + d > 0))
t0 = time.time()
total = bad = 0
for seed in seeds:
    rng = np.random.default_rng(seed)
    lens = edges + [int(x) for x in rng.integers(1, 70000, count)]
    fq = bytes(synth.fastq_like(1 << 20, seed=seed))
    tx = bytes(synth.text_like(1 << 20, seed=seed + 1000))
    fz = b"".join(hdtest.corpus_fuzz(seed, 200))
    blocks = []
    for i, n in enumerate(lens):
        kind = i % 4
        if kind == 3:
            blocks.append(bytes(rng.integers(0, 256, n, dtype=np.uint8)))
        else:
            src = (fq, tx, fz)[kind]
            o = int(rng.integers(0, max(1, len(src) - n)))
            blocks.append((src[o:o + n] * (n // max(1, len(src[o:o + n])) + 1))[:n])
    blob, offs = bytearray(), []
    for i, b in enumerate(blocks):
        blob += bytes((i * 5) % 16 if i % 3 == 0 else -len(blob) % 16)      # every third start is unaligned
        offs.append(len(blob))
        blob += b
    blob, ln = bytes(blob), [len(b) for b in blocks]
    for level in (1, 2, 3, 4, 5, 6, 9):
        for frame, twin_fn in ((pkg.FRAME_RAW | pkg.FRAME_LATENCY, hdtest.codec_twin),
                               (pkg.FRAME_RAW_FLUSH | pkg.FRAME_LATENCY, hdtest.codec_twin_flush)):
            slot = int(pkg.lib().hipdeflate_bound(max(ln), level))
            members, crc, st = pkg.batch_deflate(blob, offs, ln, level, frame, slot=slot)
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
                    print("MISMATCH seed %d block %d len %d level %d frame %#x st %d" % (seed, i, len(b), level, frame, st[i]), flush=True)
            # Round 5: levels >= 3 in latency mode are the workgroup parse + the workgroup emit kernel, and the launch's shape
            # decides the schedule -- blocks up to 64 KiB only: k_emit_wg; <= 64 blocks: the parse of a block shared by four
            # workgroups, <= 128: by two -- never the bytes: the same blocks again in launches of 48 and of 100
            if level >= 3:
                small = [i for i, b in enumerate(blocks) if len(b) <= 65536]
                for chunk in (48, 100):
                    for c0 in range(0, len(small), chunk):
                        idx = small[c0:c0 + chunk]
                        sub, so, sl = bytearray(), [], []
                        for i in idx:
                            sub += bytes(-len(sub) % 16)
                            so.append(len(sub))
                            sl.append(len(blocks[i]))
                            sub += blocks[i]
                        m2, c2, s2 = pkg.batch_deflate(bytes(sub), so, sl, level, frame, slot=slot)
                        for j, i in enumerate(idx):
                            total += 1
                            if not (s2[j] == 0 and m2[j] == twins[i][1] and int(c2[j]) == zlib.crc32(blocks[i])):
                                bad += 1
                                print("MISMATCH (launch of %d) seed %d block %d len %d level %d frame %#x st %d" % (chunk, seed, i, len(blocks[i]), level, frame, s2[j]), flush=True)
        print("seed %d level %d done, %d comparisons so far, %d bad, %.0f s" % (seed, level, total, bad, time.time() - t0), flush=True)
stalls = int(pkg.lib().hipdeflate_stall_count())
print("BIG_FUZZ_LAT %s: %d comparisons, %d bad, %d stalls" % ("OK" if bad == 0 and stalls == 0 else "FAILED", total, bad, stalls))
sys.exit(1 if bad or stalls else 0)
