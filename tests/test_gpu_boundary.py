"""GPU tests of the drop-in boundary as the reference USES it (VERDICT r3 "next" 1):

  * hip_inflate from many threads at once -- `7bgzf -d -@N` is a thread per block through zlibutil_auto_inflate
    (applet/7bgzf.c:330-345, lib/zlibutil.c:82-93): concurrent calls share launches of the latency kernel
    (k_inflate_lat) and must give, call by call, the verdicts and bytes of the batch decoder (which the parity
    suite pins to libdeflate's);
  * the device list (`hipdeflate_init(devices...)`): HIPDEFLATE_DEVICES=0,0 makes two independent contexts on the one
    card of the box; every entry gives the same bytes, pipes / latency contexts / the per-block codecs work on either,
    and `hd7bgzf -g 2` writes byte for byte what `-g 1` writes.
"""
import base64
import json
import os
import subprocess
import sys
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import hdtest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = hdtest.pkg()
    assert os.path.exists(p.LIB_PATH), "libhipdeflate.so missing: run __graft_entry__.build()"
    assert p.available(), "no usable MI355X: the HIP path must be the one that runs"
    return p


def load(name):
    return json.load(open(os.path.join(hdtest.GOLDEN, name)))


def test_hip_inflate_from_16_threads_gives_the_batch_decoders_verdicts(pkg):
    """reference-encoded streams (152), the 151 malformed igzip vectors and 600 mutants with libdeflate's verdicts, each
    through its own hip_inflate call from a pool of 16 threads (ctypes drops the GIL around the call): status and bytes
    equal the batch decoder's, which test_gpu_parity.py pins to the oracle and the golden SHA-256s."""
    zs, caps, want_sha = [], [], []
    for s in load("ref_streams.json"):
        zs.append(base64.b64decode(s["stream"]) + b"\xaa" * 8)
        caps.append(s["out_len"])
        want_sha.append(s["out_sha256"])
    for v in load("inflate_std_vects.json"):
        zs.append(base64.b64decode(v["data"]))
        caps.append(1 << 16)
        want_sha.append(None)
    for m in load("mutants.json"):
        zs.append(base64.b64decode(m["stream"]))
        caps.append(m["cap"])
        want_sha.append(m["out_sha256"] if m["libdeflate"] == 0 else None)
    outs, _, st = pkg.batch_inflate(zs, caps)

    def one(i):
        return pkg.hip_inflate(zs[i], caps[i])
    with ThreadPoolExecutor(16) as ex:
        got = list(ex.map(one, range(len(zs))))
    accepted = 0
    for i, (r, out) in enumerate(got):
        assert r == int(st[i]), (i, r, int(st[i]))
        if r == 0:
            accepted += 1
            assert out == outs[i], i
            if want_sha[i]:
                assert hdtest.sha(out) == want_sha[i]
    assert accepted > 250
    # the oracle's word on the malformed vectors once more, call by call
    for i in range(152, 152 + 151):
        assert got[i][0] != 0 and hdtest.oracle_inflate(zs[i], caps[i])[0] != 0


@pytest.mark.timeout(120)
def test_hip_inflate_runs_of_long_codewords_between_windows(pkg):
    """The per-call kernel's wavefronts talk through rings whose tails move with the WINDOWS (hd_inflate_lat.hpp): a run of
    tokens that only the scalar path takes -- codewords longer than the direct table's 9 bits, one after the other -- moves the
    bit position on with no window in between, and the spec wavefront must still reach the chunk the next window needs (round 5:
    the tail is moved by the front when the sort wavefront is through; before that this stream hung the kernel).  Streams: four
    frequent bytes and 250 bytes that occur once in a row (Huffman-only: every one of them a 13..15-bit literal), runs of them at
    the start, in the middle, at the end and back to back; through hip_inflate (a lone call and 8 threads) and the batch decoder."""
    rare = bytes(range(5, 255))
    body = b"abcd" * 6000
    cases = [body + rare + body, rare + body, body + rare, body + rare + rare[::-1] + rare + body, rare * 3,
             body[:999] + rare + body[:57] + rare[::-1] + body[:3] + rare + body]
    zs = []
    for d in cases:
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_HUFFMAN_ONLY)
        zs.append(co.compress(d) + co.flush())
    for d, z in zip(cases, zs):
        r, out = pkg.hip_inflate(z, len(d))
        assert r == 0 and out == d
        assert pkg.batch_inflate([z], [len(d)])[0][0] == d
    with ThreadPoolExecutor(8) as ex:
        for k, (r, out) in enumerate(ex.map(lambda k: pkg.hip_inflate(zs[k % len(zs)], len(cases[k % len(zs)])), range(64))):
            assert r == 0 and out == cases[k % len(zs)]


def test_hip_inflate_threads_full_blocks_flush_form_and_oversize(pkg):
    """64 threads x full 0xff00-byte blocks (zlib 1/6/9, our levels 1 and 6), the flush form side by side with the final
    form (separate batches), one byte of room less (3 = INSUFFICIENT_SPACE), and a 1 MiB member that is larger than a
    batch's arena (goes alone)."""
    s = hdtest.synth()
    blocks = [bytes(s.fastq_like(0xff00, seed=40 + k)) for k in range(6)] + [bytes(s.text_like(0xff00, seed=50 + k)) for k in range(6)]
    jobs = []
    for k, b in enumerate(blocks):
        c = zlib.compressobj([1, 6, 9][k % 3], zlib.DEFLATED, -15)
        jobs.append(("final", c.compress(b) + c.flush(), b))
        jobs.append(("final", pkg.hip_deflate(b, 1 if k & 1 else 6)[1], b))
        jobs.append(("flush", pkg.hip_deflate_flush(b[:30000], 1 + k % 3)[1], b[:30000]))
    jobs = jobs * 6

    def one(j):
        kind, z, want = j
        f = pkg.hip_inflate if kind == "final" else pkg.hip_inflate_flush
        r, out = f(z + b"\x00" * 8 if kind == "final" else z, len(want))
        r2, _ = f(z, len(want) - 1)
        return r == 0 and out == want and r2 == 3
    with ThreadPoolExecutor(64) as ex:
        assert all(ex.map(one, jobs))
    big = bytes(s.text_like(1 << 20, seed=9))
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    zbig = c.compress(big) + c.flush()
    with ThreadPoolExecutor(4) as ex:
        for r, out in ex.map(lambda _: pkg.hip_inflate(zbig, len(big)), range(4)):
            assert r == 0 and out == big


CHILD = r'''
import ctypes, importlib, os, sys, zlib
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
pkg = importlib.import_module("7bgzf_amd")
synth = importlib.import_module("7bgzf_amd.synth")
L = pkg.lib()
devs = (ctypes.c_int * 2)(0, 0)
assert L.hipdeflate_init_devices(devs, 2) == 0
assert L.hipdeflate_device_count() == 2
assert L.hipdeflate_init_devices(devs, 2) == 0                      # the same list again: fine
assert L.hipdeflate_init_devices((ctypes.c_int * 1)(0), 1) == pkg.HD_E_ARG    # another list: refused
assert L.hipdeflate_use_device(2) == pkg.HD_E_ARG
data = bytes(synth.fastq_like(40 * 0xff00 + 99, seed=8))
res = {}
for idx in (0, 1):
    assert L.hipdeflate_use_device(idx) == 0
    for level in (1, 2, 6):
        res[idx, level] = pkg.bgzf_compress_bytes(data, level)
        assert pkg.bgzf_decompress_bytes(res[idx, level]) == data
    res[idx, "pipe"] = pkg.pipe_compress(data, 6, per_batch=8)          # opened on the thread's entry
    res[idx, "codec"] = pkg.hip_deflate(data[:0xff00], 2)[1]
assert all(res[0, k] == res[1, k] for k in (1, 2, 6, "pipe", "codec"))
assert res[0, "pipe"] + pkg.BGZF_EOF == res[0, 6]
# per-block decoders from 32 threads: their batches alternate between the two entries
z = zlib.compress(data[:0xff00], 6)[2:-4]
with ThreadPoolExecutor(32) as ex:
    for r, out in ex.map(lambda _: pkg.hip_inflate(z, 0xff00), range(256)):
        assert r == 0 and out == data[:0xff00]
# a latency context on entry 1 while the thread sits on entry 0
assert L.hipdeflate_use_device(0) == 0
c = L.hipdeflate_lat_open_on(1, 1, pkg.FRAME_BGZF | pkg.FRAME_LATENCY, 4, 0xff00)
assert c
blk = np.frombuffer(data[:0xff00], dtype=np.uint8)
ctypes.memmove(L.hipdeflate_lat_input(c, 0), blk.ctypes.data, 0xff00)
lens = (ctypes.c_uint32 * 1)(0xff00)
assert L.hipdeflate_lat_run(c, lens, 1) == 0
olen, st = ctypes.c_uint32(), ctypes.c_int32()
m = L.hipdeflate_lat_output(c, 0, ctypes.byref(olen), None, ctypes.byref(st))
member = ctypes.string_at(m, olen.value)
assert st.value == 0 and pkg.bgzf_decompress_bytes(member + pkg.BGZF_EOF) == data[:0xff00]
L.hipdeflate_lat_close(c)
L.hipdeflate_shutdown()
print("DEVICE-LIST-OK", L.hipdeflate_device_count())
'''


def test_device_list_two_contexts_on_one_card(pkg):
    """hipdeflate_init_devices([0, 0]) in a fresh process (the list is fixed by the first use): both entries give the
    same bytes at levels 1 / 2 / 6, through the batch API, a pipe, the per-block codecs; 32 threads of hip_inflate;
    a latency context opened on the other entry; a different list is refused."""
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": hdtest.ROOT}], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "DEVICE-LIST-OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])


def test_hd7bgzf_g2_over_the_device_list_equals_g1(pkg, tmp_path):
    """hd7bgzf -g 2 with HIPDEFLATE_DEVICES=0,0 (two contexts on the box's one card): stdin filter and file-to-file path
    write byte for byte the stream of -g 1, at level 1 and level 6; -d -g 2 reads it back."""
    exe = os.path.join(os.path.dirname(pkg.LIB_PATH), "hd7bgzf")
    data = bytes(hdtest.synth().fastq_like(9 * 64 * 0xff00 + 4321, seed=12))
    src = str(tmp_path / "in.bin")
    open(src, "wb").write(data)
    env2 = dict(os.environ, HIPDEFLATE_DEVICES="0,0", HD7BGZF_BATCH="64")
    env1 = dict(os.environ, HD7BGZF_BATCH="64")
    for level in (1, 6):
        outs = []
        for g, env in ((1, env1), (2, env2)):
            o = str(tmp_path / ("o%d_%d.bgz" % (level, g)))
            p = subprocess.run([exe, "-G%d" % level, "-g%d" % g, "-@4", "-i", src, "-o", o], env=env, capture_output=True, text=True)
            assert p.returncode == 0, p.stderr[-2000:]
            outs.append(open(o, "rb").read())
            p = subprocess.run([exe, "-G%d" % level, "-g%d" % g], stdin=open(src, "rb"), env=env, capture_output=True)
            assert p.returncode == 0, p.stderr[-2000:]
            outs.append(p.stdout)
        assert outs[0] == outs[1] == outs[2] == outs[3], level
        assert pkg.bgzf_decompress_bytes(outs[0]) == data
        p = subprocess.run([exe, "-d", "-g2"], input=outs[0], env=env2, capture_output=True)
        assert p.returncode == 0 and p.stdout == data
    # -g N needs N entries in the list: a shorter HIPDEFLATE_DEVICES is an error, not N pipes on one entry
    p = subprocess.run([exe, "-G1", "-g3"], input=data[:100000], env=env2, capture_output=True)
    assert p.returncode != 0 and b"has 2 entries" in p.stderr, p.stderr[-500:]


def test_hip_deflate_from_32_threads_mixed_levels_frames_and_rooms(pkg):
    """hip_deflate / hip_deflate_flush from 32 threads at once, every thread with its own levels, frames, lengths and rooms:
    callers whose room makes the bytes independent of it share launches (bgzf_hook.c hd_codec_batch, an engine per level and
    frame), the others -- a room below the latency form's worst case or below the stored form -- take a context of their
    own; either way a call gives what the twin gives for that block and that room (lib/zlibutil.h:47: re-entrant, no shared
    state the caller could see)."""
    import threading
    import zlib
    import numpy as np
    s = hdtest.synth()
    pool = bytes(s.fastq_like(400000, seed=71)) + bytes(s.text_like(400000, seed=72)) + bytes(s.random_bytes(70000, seed=73))
    bad = []

    def work(t):
        rng = np.random.default_rng(1000 + t)
        for k in range(6):
            level = int(rng.choice([1, 2, 3, 6]))
            flush = bool(rng.integers(0, 2))
            n = int(rng.choice([1, 100, 4080, 8160, 8161, 30000, 0xff00])) if k % 2 else int(rng.integers(1, 0xff00 + 1))
            o = int(rng.integers(0, len(pool) - n))
            data = pool[o:o + n]
            cap = [n + n // 2 + 64, n + 12, max(16, n // 2), 65536][int(rng.integers(0, 4))]
            fn, tw = (pkg.hip_deflate_flush, hdtest.codec_twin_flush) if flush else (pkg.hip_deflate, hdtest.codec_twin)
            r, z = fn(data, level, cap=cap)
            rt, zt = tw(data, level, cap=cap)
            if (r != 0) != (rt != 0) or (r == 0 and (z != zt or zlib.decompressobj(-15).decompress(z + (b"\x03\x00" if flush else b"")) != data)):
                bad.append((t, k, level, flush, n, cap, r, rt))

    th = [threading.Thread(target=work, args=(t,)) for t in range(32)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad[:5]
