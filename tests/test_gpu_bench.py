"""The driver's bench contract, exercised small on the GPU box: bench.py as a child process prints ONE JSON line with the
fields the contract names; with HD_BENCH_FORCE_DIST=1 the world-1 run goes through the N > 1 path (exchange of totals,
scan with this rank's base, gather into the rank's span) on the software-pipelined passes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra_env, *args):
    env = dict(os.environ, **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gib", "1", "--tile-mib", "16", "--steps", "3",
                          "--warmup", "1", "--no-cpu", "--no-extra", *args],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_contract_small():
    j = run_bench({})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["unit"] == "GB/s" and j["dtype"] == "u8" and j["scaling"] == "weak"
    assert j["value"] > 10 and 0 < j["roofline"]["frac"] < 1 and j["roofline"]["bound"] == "hbm"
    assert abs(j["roofline"]["achieved"] / j["roofline"]["peak"] - j["roofline"]["frac"]) < 1e-3
    assert "workload" in j["config"] and 0.3 < j["config"]["ratio"] < 0.6


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, 6])
def test_bench_distributed_path_at_world_one(level):
    env = {"HD_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29541 + level), "RANK": "0",
           "WORLD_SIZE": "1", "LOCAL_RANK": "0"}
    j = run_bench(env, "--level", str(level))
    so = j["config"]["stream_offsets"]
    assert so["bases"] == [0] and so["stream_bytes"] == so["totals"][0] > 0
    assert "all_gather" in j["config"]["step"]
