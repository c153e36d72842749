"""child of tools/exp_inflate_stats.sh: runs bench.py's decode leg in-process on 1 GiB and prints the counters"""
import ctypes, importlib, sys, runpy
sys.path.insert(0, '.')
stream = sys.argv[1]
sys.argv = ["bench.py", "--no-cpu", "--mode", "decode", "--stream", stream, "--gib", "1", "--steps", "1", "--warmup", "0"] + \
    (["--level", "1"] if stream == "own" else [])
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pkg = importlib.import_module("7bgzf_amd")
out = (ctypes.c_uint64 * 8)()
pkg.lib().hipdeflate_test_inflate_stats(out)
names = ["windows", "window_tokens", "scalar_tokens", "slow_litlen", "slow_dist", "eob", "window_empty", "-"]
print({n: int(v) for n, v in zip(names, out)})
cyc = (ctypes.c_uint64 * 8)()
pkg.lib().hipdeflate_test_inflate_cycles(cyc)
tot = max(int(cyc[3]), 1)
print({"cycles": {"headers+tables": round(int(cyc[0]) / tot, 3), "windows": round(int(cyc[1]) / tot, 3),
                   "scalar token path": round(int(cyc[2]) / tot, 3), "whole kernel (sum over waves)": tot}})
m2 = (ctypes.c_uint64 * 8)()
if hasattr(pkg.lib(), "hipdeflate_test_inflate_stats2"):
    pkg.lib().hipdeflate_test_inflate_stats2(m2)
    names2 = ["matches", "groups_of_8", "groups_of_16", "simple_scalar", "far_scalar", "general", "-", "-"]
    print({"window matches": {n: int(v) for n, v in zip(names2, m2) if n != "-"}})
