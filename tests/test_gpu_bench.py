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
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):           # a launcher around pytest must not leak into the child
        if k not in extra_env:
            env.pop(k, None)
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


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """--gpus N is not a dead argument: under a launcher that started a different number of ranks bench.py stops
    before it touches the GPU (no GPU needed for this test)"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu", "--no-extra"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=1" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_bench_launcher_path_at_one_rank():
    """`python bench.py` with no launcher around it starts its ranks itself, as a child process: at N = 1 (forced by
    HD_BENCH_LAUNCH=1) the child is one torch.distributed.run rank on RCCL and the line comes back through the parent"""
    j = run_bench({"HD_BENCH_LAUNCH": "1"})
    assert j["n_gpus"] == 1 and j["ranks_seen"] == 1 and len(j["value_per_rank"]) == 1
    so = j["config"]["stream_offsets"]
    assert so["bases"] == [0] and so["first_member_offset"] == [0] and so["stream_bytes"] == so["totals"][0] > 0
    assert "nccl" in j["config"]["collective"]


@pytest.mark.gpu
def test_bench_two_ranks_share_the_card_and_exchange_totals():
    """The pipelined step with a REAL second rank on the one-GPU box: `bench.py --gpus 2` starts two ranks, both run their
    kernels on cuda:0 (HD_BENCH_DEVICE=0) over 1 GiB each and exchange the per-rank totals over gloo on CPU tensors
    (RCCL wants a device per rank).  Rank 1's span starts where rank 0's ends: its first member's offset in the whole
    stream == its base != 0 (asserted inside every rank too, bench.py encode())."""
    j = run_bench({"HD_BENCH_DIST_BACKEND": "gloo", "HD_BENCH_DEVICE": "0"}, "--gpus", "2")
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and len(j["value_per_rank"]) == 2
    so = j["config"]["stream_offsets"]
    assert len(so["totals"]) == 2 and min(so["totals"]) > 0
    assert so["bases"] == [0, so["totals"][0]] and so["first_member_offset"] == so["bases"]
    assert so["stream_bytes"] == sum(so["totals"])
    assert so["totals"][0] != so["totals"][1]                # every rank has its own seed: its own shard of the stream
    assert "gloo" in j["config"]["collective"] and "all_gather" in j["config"]["step"]
    assert abs(j["value"] - 2 * (1 << 30) * j["steps"] / (j["ms_per_step"] * j["steps"] / 1e3) / 1e9) < 0.02 * j["value"]
