"""One-off wide run on the GPU box over the BENCH data kinds: FASTQ-like and enwik-like text from many seeds, cut in
0xff00-byte BGZF blocks and 1 MiB MiGz blocks, every level 1..9, plain and latency frames: kernel bytes == twin bytes,
zlib reads them back.  usage: [HD_FUZZ_BLOCKS=65280,1048576] [HD_FUZZ_LEVELS=3,6] python tools/big_fuzz_synth.py [MiB_per_seed] [seeds...]"""
import importlib
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hdtest  # noqa: E402

pkg = importlib.import_module("7bgzf_amd")
synth = importlib.import_module("7bgzf_amd.synth")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seeds = [int(a) for a in sys.argv[2:]] or [101, 102, 103]
t0 = time.time()
total = bad = 0
for seed in seeds:
    for kind, gen in (("fastq", synth.fastq_like), ("text", synth.text_like)):
        data = gen(mib << 20, seed=seed).tobytes()
        for bs in [int(x) for x in os.environ.get("HD_FUZZ_BLOCKS", "65280,1048576").split(",")]:
            offs = list(range(0, len(data), bs))
            lens = [min(bs, len(data) - o) for o in offs]
            blocks = [data[o:o + n] for o, n in zip(offs, lens)]
            for level in [int(x) for x in os.environ.get("HD_FUZZ_LEVELS", "1,2,3,4,5,6,7,8,9").split(",")]:
                frames = [(pkg.FRAME_RAW, hdtest.oracle_twin)]
                if bs <= 0x10000:
                    frames.append((pkg.FRAME_RAW | pkg.FRAME_LATENCY, hdtest.codec_twin))
                for frame, twin_fn in frames:
                    slot = int(pkg.lib().hipdeflate_bound(max(lens), level))
                    members, crc, st = pkg.batch_deflate(data, offs, lens, level, frame, slot=slot)
                    with ThreadPoolExecutor(min(64, os.cpu_count() or 16)) as ex:
                        twins = list(ex.map(lambda b: twin_fn(b, level, cap=slot), blocks))
                    for i, b in enumerate(blocks):
                        total += 1
                        ok = st[i] == 0 and twins[i][0] == 0 and members[i] == twins[i][1] and int(crc[i]) == zlib.crc32(b)
                        ok = ok and zlib.decompress(members[i], -15) == b
                        if not ok:
                            bad += 1
                            print("MISMATCH seed %d %s block %d size %d level %d frame %d" % (seed, kind, i, bs, level, frame), flush=True)
            print("seed %d %s block size %d done, %d comparisons so far, %d bad, %.0f s" % (seed, kind, bs, total, bad, time.time() - t0), flush=True)
stalls = int(pkg.lib().hipdeflate_stall_count())
print("BIG_FUZZ_SYNTH %s: %d comparisons, %d bad, %d stalls" % ("OK" if bad == 0 and stalls == 0 else "FAILED", total, bad, stalls))
sys.exit(1 if bad or stalls else 0)
