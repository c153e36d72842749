"""experiment: cycles of k_emit_wg (hd_emit_wg.hpp: the member of a lone block written by a workgroup) by phase, seen from
wavefront 0; needs a library built with make -C 7bgzf_amd/csrc EXTRA=-DHD_EMIT_STATS.  usage: python tools/exp_emit_wg_stats.py [level] [fastq|text]"""
import ctypes, importlib, sys
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module("7bgzf_amd")
synth = importlib.import_module("7bgzf_amd.synth")
level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
kind = sys.argv[2] if len(sys.argv) > 2 else "fastq"
nb, B = 16, 0xff00
data = (synth.fastq_like if kind == "fastq" else synth.text_like)(nb * B, seed=1234)
offs = np.arange(nb, dtype=np.uint64) * B
lens = np.full(nb, B, dtype=np.uint32)
out = (ctypes.c_uint64 * 32)()
pkg.batch_deflate(data, offs, lens, level, pkg.FRAME_BGZF | pkg.FRAME_LATENCY, slot=65536)
pkg.lib().hipdeflate_test_emit_stats32(out)
v0 = [int(x) for x in out]
reps = 50
for _ in range(reps):
    pkg.batch_deflate(data, offs, lens, level, pkg.FRAME_BGZF | pkg.FRAME_LATENCY, slot=65536)
pkg.lib().hipdeflate_test_emit_stats32(out)
v = [(int(x) - y) / reps / nb for x, y in zip(out, v0)]       # cycles per member (wavefront 0's clock)
names = ["init", "cut scan (w15) / pieces' symbol counts", "blocks' symbol sums", "codes, rle, precode, costs, table (a wavefront per block)",
         "headers / pieces' weights", "token coding", "(barrier)", "trailer + copy out"]
print({"level": level, "data": kind, "cycles_per_member": {n: round(x) for n, x in zip(names, v[:8])}, "sum": round(sum(v[:8])),
       "build_code_litlen_cumulative": dict(zip(["copy+keys", "rank sort", "merge+depths+leaf levels", "overflow+first codes", "lengths", "codewords"],
                                                 [round(x) for x in v[8:14]])),
       "builder_wave0": dict(zip(["litlen code", "(offset code: beside it)", "rle", "precode", "costs", "static? + table"], [round(x) for x in v[16:22]]))})
