"""child of tools/clock_stamps.sh: one bench.py leg in-process on a -DHD_CLOCK_STAMPS library, then the in-kernel clock of
every stamped kernel = sum of delta s_memtime / sum of delta s_memrealtime x 100 MHz over all its waves"""
import ctypes, importlib, json, sys, runpy, io, contextlib
sys.path.insert(0, '.')
name = sys.argv[1]
sys.argv = ["bench.py", "--no-cpu", "--no-extra"] + sys.argv[2:]
buf = io.StringIO()
try:
    with contextlib.redirect_stdout(buf):
        runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
line = json.loads(buf.getvalue().strip().splitlines()[-1])
pkg = importlib.import_module("7bgzf_amd")
out = (ctypes.c_uint64 * 16)()
assert pkg.lib().hipdeflate_test_clock(out) == 0
kern = ["k_deflate_static (level 1)", "k_deflate_dynamic", "k_inflate", "k_deflate_static<TOK> (parse)"]
res = {"leg": name, "bench_value_GBps": line["value"], "kernel_ms_avg": line["roofline"]["kernel_ms_avg"], "kernels": {}}
for i, k in enumerate(kern):
    cyc, ticks, waves = int(out[4 * i]), int(out[4 * i + 1]), int(out[4 * i + 2])
    if waves:
        res["kernels"][k] = {"clock_mhz": round(cyc / ticks * 100.0, 1), "waves": waves,
                             "mean_wave_cycles": round(cyc / waves), "mean_wave_us": round(ticks / waves / 100.0, 2)}
print(json.dumps(res))
