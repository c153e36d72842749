"""bench.py with the number of emit wavefronts a CU keeps beside the parse set first (hipdeflate_test_beside): python3 tools/bench_keep.py KEEP [bench args]"""
import importlib, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
keep = int(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
importlib.import_module("7bgzf_amd").lib().hipdeflate_test_beside(keep, 0)
runpy.run_path(sys.argv[0], run_name="__main__")
