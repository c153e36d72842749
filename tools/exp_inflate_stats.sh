# experiment: where the inflate kernel's tokens go (windows / scalar loop, by cause).  Needs a library built
# with -DHD_INFLATE_STATS:  make -C 7bgzf_amd/csrc clean && make -C 7bgzf_amd/csrc EXTRA=-DHD_INFLATE_STATS
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
python3 - <<'PY'
import ctypes, importlib, subprocess, sys, json
sys.path.insert(0, '.')
for stream in ("libdeflate6", "zlib6", "own"):
    p = subprocess.run([sys.executable, "tools/exp_inflate_stats_child.py", stream], capture_output=True, text=True)
    print(stream, "\n".join(p.stdout.strip().splitlines()[-3:]) if p.stdout.strip() else p.stderr[-400:])
PY
