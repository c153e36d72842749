"""experiment: mean lifetime of a wavefront of the latency-mode kernels (a -DHD_CLOCK_STAMPS library: make -C 7bgzf_amd/csrc
EXTRA=-DHD_CLOCK_STAMPS) against the kernels' durations in a trace -- what of a latency batch is NOT the waves' own work.
usage: python tools/clock_stamps_lat.py [level] [blocks] [fastq|random|repeat]"""
import ctypes, importlib, json, sys
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module("7bgzf_amd")
synth = importlib.import_module("7bgzf_amd.synth")
level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = 0xff00
kind = sys.argv[3] if len(sys.argv) > 3 else "fastq"
data = (np.random.default_rng(7).integers(0, 256, nb * B, dtype=np.uint8) if kind == "random" else
        np.tile(synth.fastq_like(2048, seed=1234), nb * B // 2048 + 1)[:nb * B].copy() if kind == "repeat" else
        synth.fastq_like(nb * B, seed=1234))
offs = np.arange(nb, dtype=np.uint64) * B
lens = np.full(nb, B, dtype=np.uint32)
out = (ctypes.c_uint64 * 16)()
for _ in range(3):
    pkg.batch_deflate(data, offs, lens, level, pkg.FRAME_BGZF | pkg.FRAME_LATENCY, slot=65536)
pkg.lib().hipdeflate_test_clock(out)                     # reads and resets
marks = (ctypes.c_uint64 * 16)()
pkg.lib().hipdeflate_test_clock_marks(marks)
for _ in range(50):
    pkg.batch_deflate(data, offs, lens, level, pkg.FRAME_BGZF | pkg.FRAME_LATENCY, slot=65536)
assert pkg.lib().hipdeflate_test_clock(out) == 0
kern = ["k_deflate_static (level 1)", "k_deflate_dynamic", "k_inflate", "k_deflate_static<TOK> (parse)"]
res = {"level": level, "blocks": nb, "data": kind, "kernels": {}}
for i, k in enumerate(kern):
    cyc, ticks, waves = int(out[4 * i]), int(out[4 * i + 1]), int(out[4 * i + 2])
    if waves:
        res["kernels"][k] = {"clock_mhz": round(cyc / ticks * 100.0, 1), "waves_per_launch": waves / 50,
                             "mean_wave_cycles": round(cyc / waves), "mean_wave_us": round(ticks / waves / 100.0, 2)}
pkg.lib().hipdeflate_test_clock_marks(marks)
res["static_or_parse_wave_marks_us"] = {k: [round(int(marks[i]) / max(1, int(marks[8 + i])) / 2380.0, 2), int(marks[8 + i]) // 50]
                                        for i, k in enumerate(["prologue_done", "steps_done", "crc_done"])}
print(json.dumps(res))
