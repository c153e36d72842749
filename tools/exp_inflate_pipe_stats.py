"""experiment: where the two wavefronts of k_inflate_lat (hd_inflate_lat.hpp) spend a lone hip_inflate call -- each one's cycles
in all and those it waited for the other; needs a library built with make -C 7bgzf_amd/csrc EXTRA=-DHD_INFLATE_STATS.
usage: python tools/exp_inflate_pipe_stats.py"""
import ctypes, importlib, sys, zlib
sys.path.insert(0, '.')
pkg = importlib.import_module("7bgzf_amd")
synth = importlib.import_module("7bgzf_amd.synth")
blk = bytes(synth.fastq_like(0xff00, seed=5))
for name, z in (("zlib6", zlib.compress(blk, 6)[2:-4]), ("own_level1", pkg.hip_deflate(blk, 1)[1])):
    a, b, c = (ctypes.c_uint64 * 8)(), (ctypes.c_uint64 * 8)(), (ctypes.c_uint64 * 8)()
    pkg.lib().hipdeflate_test_inflate_stats(a); pkg.lib().hipdeflate_test_inflate_stats2(b); pkg.lib().hipdeflate_test_inflate_cycles(c)
    t0, w0, c0 = [int(x) for x in a], [int(x) for x in b], [int(x) for x in c]
    reps = 100
    for _ in range(reps):
        r, back = pkg.hip_inflate(z, 0xff00)
        assert r == 0 and back == blk
    pkg.lib().hipdeflate_test_inflate_stats(a); pkg.lib().hipdeflate_test_inflate_stats2(b); pkg.lib().hipdeflate_test_inflate_cycles(c)
    t, w = [(int(x) - y) / reps for x, y in zip(a, t0)], [(int(x) - y) / reps for x, y in zip(b, w0)]
    ph = [(int(x) - y) / reps for x, y in zip(c, c0)]
    print({"stream": name, "cycles_per_call": {"front": round(t[6]), "front waiting for a free record": round(w[6]),
                                               "back": round(t[7]), "back waiting for a record": round(w[7])},
           "front_window_loop": dict(zip(["wait for the spec words + fields", "walk", "prefix sum + budget", "distance check", "hand-over to the sort wavefront (+ slot wait)"], [round(x) for x in ph[:5]]))})
