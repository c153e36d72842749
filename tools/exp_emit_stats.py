"""experiment: cycles of the emit-only kernel by phase; needs a library built with
make -C 7bgzf_amd/csrc EXTRA=-DHD_EMIT_STATS.  usage: python tools/exp_emit_stats.py [bench args]"""
import ctypes, importlib, sys, runpy
sys.path.insert(0, '.')
extra = sys.argv[1:]
sys.argv = ["bench.py", "--no-cpu", "--no-extra", "--gib", "2", "--steps", "1", "--warmup", "0", "--level", "2"] + extra
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pkg = importlib.import_module("7bgzf_amd")
out = (ctypes.c_uint64 * 16)()
pkg.lib().hipdeflate_test_emit_stats(out)
names = ["build litlen+offset codes", "lens copy + RLE (lane 0)", "precode, costs, header", "token loop (cumulative marks)"]
v = [int(x) for x in out]
# marks 1..3 are cumulative from the same start (EMIT_T0 after the code construction)
print({"build_codes": v[0], "rle_lane0": v[1], "precode_costs_header": v[2] - v[1], "tokens": v[3] - v[2]})
# the workgroup records (levels >= 6): marks 4..6 are cumulative from the start of a DEFLATE block
if v[6]:
    print({"wg_cut_scan": v[4], "wg_symbol_count_pass": v[5] - v[4], "wg_flush_block": v[6] - v[5], "wg_total": v[6]})
