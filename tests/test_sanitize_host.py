"""SURVEY.md section 5 row 2, VERDICT r1 item 9: the C host side under sanitizers.  The pool has no GPU sanitizers,
so bgzf_hook.c (leader / member batching, spin-then-sleep waits, three contexts in flight) and zlibutil_hip.c are
compiled with -fsanitize=address,undefined and with -fsanitize=thread against tests/native/stub_hipdeflate.c (a
TEST-ONLY stand-in for the device entry points: stored members) and driven by tests/native/hook_stress.c:
64 threads, every member checked against its block (HD_SAN_FULL=1: 10,000 calls per thread, a few minutes; the
default run is sized for the CPU suite)."""
import os
import subprocess

import pytest

import hdtest

SRC = os.path.join(hdtest.ROOT, "7bgzf_amd", "csrc")
NAT = os.path.join(hdtest.ROOT, "tests", "native")
INC = os.path.join(hdtest.ROOT, "include")


def _build(tmp_path, flags, name):
    exe = str(tmp_path / name)
    cmd = ["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-std=gnu11", "-pthread", "-Wall"] + flags + [
        "-I" + INC, "-I" + SRC, "-o", exe, os.path.join(NAT, "hook_stress.c"), os.path.join(NAT, "stub_hipdeflate.c"),
        os.path.join(SRC, "bgzf_hook.c"), os.path.join(SRC, "zlibutil_hip.c")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return exe


def _run(exe, threads, calls, extra_env=None):
    env = dict(os.environ)
    env.pop("BGZF_METHOD", None)                 # unset: hip at the reference's default level (6)
    env.update(ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1", HIPDEFLATE_HOOK_STATS="1")
    env.update(extra_env or {})
    p = subprocess.run([exe, str(threads), str(calls)], capture_output=True, text=True, env=env, timeout=900)
    return p


def test_hook_and_zlibutil_mirror_under_asan_ubsan(tmp_path):
    exe = _build(tmp_path, ["-fsanitize=address,undefined"], "hook_asan")
    p = _run(exe, 64, 10000 if os.environ.get("HD_SAN_FULL") else 1500)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-3000:])
    assert "0 bad" in p.stdout and "ERROR" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    assert "hipdeflate hook (" in p.stderr                     # the batching ran (statistics line at exit)
    # small spin budget: the members' sleep / wake path
    p = _run(exe, 16, 1000, {"HIPDEFLATE_SPIN_US": "0", "HIPDEFLATE_BATCH_US": "200"})
    assert p.returncode == 0 and "0 bad" in p.stdout and "ERROR" not in p.stderr, p.stderr[-3000:]


def test_hook_under_tsan(tmp_path):
    exe = _build(tmp_path, ["-fsanitize=thread"], "hook_tsan")
    for env in ({}, {"HIPDEFLATE_SPIN_US": "0"}, {"HIPDEFLATE_BATCH_BLOCKS": "3"}):
        p = _run(exe, 64, 2000 if os.environ.get("HD_SAN_FULL") else 400, env)
        assert p.returncode == 0, (env, p.stdout[-500:], p.stderr[-3000:])
        assert "0 bad" in p.stdout and "WARNING: ThreadSanitizer" not in p.stderr, (env, p.stderr[-3000:])


def test_hook_maps_foreign_methods_to_hip_at_the_reference_level(tmp_path):
    """BGZF_METHOD=libdeflate6 left in the environment: the reference would compress with libdeflate at level 6; this
    library keeps writing with its own coder at that level and says so once (bgzf_hook.c parse_env)"""
    exe = _build(tmp_path, ["-fsanitize=address,undefined"], "hook_asan2")
    p = _run(exe, 2, 10, {"BGZF_METHOD": "libdeflate6"})
    assert p.returncode == 0 and "0 bad" in p.stdout and p.stderr.count("coding with hip6") == 1, p.stderr[-2000:]
    p = _run(exe, 2, 10, {"BGZF_METHOD": "nosuchcoder"})
    assert p.returncode == 0 and "coding with hip6" in p.stderr, p.stderr[-2000:]
