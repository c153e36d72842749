"""experiment: cycles of the emit-only kernel by phase in LATENCY mode (one wavefront per 8160-byte segment, alone on its
SIMD); needs a library built with make -C 7bgzf_amd/csrc EXTRA=-DHD_EMIT_STATS.  usage: python tools/exp_emit_stats_lat.py [level]"""
import ctypes, importlib, sys
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module("7bgzf_amd")
synth = importlib.import_module("7bgzf_amd.synth")
level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nb, B = 16, 0xff00
data = synth.fastq_like(nb * B, seed=1234)
offs = np.arange(nb, dtype=np.uint64) * B
lens = np.full(nb, B, dtype=np.uint32)
out = (ctypes.c_uint64 * 16)()
pkg.batch_deflate(data, offs, lens, level, pkg.FRAME_BGZF | pkg.FRAME_LATENCY, slot=65536)
pkg.lib().hipdeflate_test_emit_stats(out)
v0 = [int(x) for x in out]
reps = 50
for _ in range(reps):
    pkg.batch_deflate(data, offs, lens, level, pkg.FRAME_BGZF | pkg.FRAME_LATENCY, slot=65536)
pkg.lib().hipdeflate_test_emit_stats(out)
v = [(int(x) - y) / reps / (nb * 8) for x, y in zip(out, v0)]       # cycles per emit wave (8 segments per block)
# marks 1..3 are cumulative from the same start (EMIT_T0 after the code construction)
print({"level": level, "cycles_per_segment": {"build_codes": round(v[0]), "lens_rle": round(v[1]), "precode_costs_header": round(v[2] - v[1]),
       "tokens": round(v[3] - v[2])},
       "build_code_litlen_cumulative": dict(zip(["copy+keys", "rank sort", "merge+depths+leaf levels", "overflow+first codes", "lengths", "codewords"],
                                                 [round(x) for x in v[8:14]]))})
