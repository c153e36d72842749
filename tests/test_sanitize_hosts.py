"""the seven container hosts' READERS under AddressSanitizer + UBSan with a mutational fuzz (CPU box).

hd_{bgzf,dictzip,razf,gzinga,ciso,daxcr,png}_host.c parse untrusted offset tables, member lengths, index members and PNG
chunk lengths (roles of applet/7bgzf.c:295-365, 7dictzip.c:318-323, 7razf.c, 7gzinga.c:76-214, 7ciso.c:87, 7daxcr.c:130,
7png.c:296-331).  Each host is compiled with -fsanitize=address,undefined against tests/native/stub_containers.c (a
TEST-ONLY stand-in for the device entry points: stored blocks, inflate of stored blocks) and driven by
tests/native/container_fuzz.c: the host writes a seed container from seeded data, then N mutants of it go through the
reader in forked children.  Contract: it exits -- never a sanitizer report, a signal or a hang.  Default 400 mutants per
container (the CPU suite's budget); HD_SAN_FULL=1 runs 2,000."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import hdtest

SRC = os.path.join(hdtest.ROOT, "7bgzf_amd", "csrc")
NAT = os.path.join(hdtest.ROOT, "tests", "native")
INC = os.path.join(hdtest.ROOT, "include")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:exitcode=99:abort_on_error=0:allocator_may_return_null=1",
           UBSAN_OPTIONS="exitcode=99:halt_on_error=1:print_stacktrace=1")


def build(tmp_path, host, fuzz):
    exe = str(tmp_path / ("%s_%s" % (host, "fuzz" if fuzz else "cli")))
    src = [os.path.join(SRC, "hd_%s_host.c" % host), os.path.join(NAT, "stub_containers.c")]
    if host == "png":
        src.append(os.path.join(SRC, "zlibutil_hip.c"))
    cc = ["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-std=gnu11", "-pthread", "-Wall"] + SAN + ["-I" + INC, "-I" + SRC]
    if fuzz:
        # the host's own main() becomes host_main (that one translation unit only) and the fuzz driver calls it
        obj = exe + "_host.o"
        p = subprocess.run(cc + ["-Dmain=host_main", "-c", src[0], "-o", obj], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-3000:]
        src = [obj] + src[1:] + [os.path.join(NAT, "container_fuzz.c")]
    p = subprocess.run(cc + ["-o", exe] + src, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return exe


def tiny_png(w=64, h=48, seed=5):
    """a valid 8-bit RGB PNG whose IDAT is zlib with STORED blocks (the stub inflates nothing else)"""
    rng = np.random.default_rng(seed)
    raw = b"".join(b"\x00" + bytes(rng.integers(0, 256, 3 * w, dtype=np.uint8)) for _ in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    idat = zlib.compress(raw, 0)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"tEXt", b"k\x00v") +
            chunk(b"IDAT", idat[:100]) + chunk(b"IDAT", idat[100:]) + chunk(b"IEND", b""))


# host -> (how the CLI writes the seed, fuzz mode, reader args)
HOSTS = {
    "bgzf":    (lambda cli, src, out: subprocess.run([cli, "-G1"], stdin=open(src, "rb"), stdout=open(out, "wb"), env=ENV), "stdin", ["-d"]),
    "dictzip": (lambda cli, src, out: subprocess.run([cli, "-G1", src, out], env=ENV), "file", ["-d", "@"]),
    "razf":    (lambda cli, src, out: subprocess.run([cli, "-G1", src], stdout=open(out, "wb"), env=ENV), "file", ["-d", "@"]),
    "gzinga":  (lambda cli, src, out: subprocess.run([cli, "-G1"], stdin=open(src, "rb"), stdout=open(out, "wb"), env=ENV), "file", ["-d", "@"]),
    "ciso":    (lambda cli, src, out: subprocess.run([cli, "-G1", src, out], env=ENV), "stdin", ["-d"]),
    "daxcr":   (lambda cli, src, out: subprocess.run([cli, "-G1", src, out], env=ENV), "stdin", ["-d"]),
    "png":     (None, "stdin", ["-G1"]),
}


@pytest.mark.parametrize("host", sorted(HOSTS))
def test_container_reader_survives_mutated_files_under_asan_ubsan(tmp_path, host):
    write, mode, args = HOSTS[host]
    seed = str(tmp_path / ("seed." + host))
    if write is None:
        open(seed, "wb").write(tiny_png())
    else:
        cli = build(tmp_path, host, fuzz=False)
        src = str(tmp_path / "plain.bin")
        # a few hundred KiB: several chunks / sectors / members and a multi-entry table in every format
        data = bytes(hdtest.synth().fastq_like(300000 if host != "ciso" else 64 * 2048, seed=11))
        open(src, "wb").write(data)
        p = write(cli, src, seed)
        assert p.returncode == 0 and os.path.getsize(seed) > len(data) // 2
    fuzz = build(tmp_path, host, fuzz=True)
    n = 2000 if os.environ.get("HD_SAN_FULL") else 400
    p = subprocess.run([fuzz, seed, str(n), "7", mode] + args, capture_output=True, text=True, env=ENV, timeout=1500)
    assert p.returncode == 0 and ("%d mutants, 0 bad" % n) in p.stdout, (p.stdout[-500:], p.stderr[-3000:])


def test_hd7bgzf_deals_batches_round_robin_to_n_devices_under_asan(tmp_path):
    """hd7bgzf -g N (SURVEY 8(b) device list; VERDICT r3 "next" 1b): one pipe per entry of the device list, batch j on pipe
    j % N, results fetched in the same order.  Host logic only -- the sanitizer build against the test stub, whose
    "devices" are all the same CPU code: the stream of -g 3 (stdin filter and file-to-file path, five small batches) is
    byte for byte that of -g 1, and -d -g 2 gives the input back."""
    cli = build(tmp_path, "bgzf", fuzz=False)
    data = bytes(hdtest.synth().fastq_like(5 * 16 * 0xff00 + 777, seed=3))
    src = str(tmp_path / "in.bin")
    open(src, "wb").write(data)
    env = dict(ENV, HD7BGZF_BATCH="16")
    outs = {}
    for g in (1, 3):
        o = str(tmp_path / ("f%d.bgz" % g))
        p = subprocess.run([cli, "-G1", "-g%d" % g, "-@2", "-i", src, "-o", o], env=env, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        outs["file", g] = open(o, "rb").read()
        p = subprocess.run([cli, "-G1", "-g%d" % g], stdin=open(src, "rb"), env=env, capture_output=True)
        assert p.returncode == 0, p.stderr[-2000:]
        outs["pipe", g] = p.stdout
    assert outs["file", 1] == outs["file", 3] == outs["pipe", 1] == outs["pipe", 3]
    assert outs["file", 1].endswith(hdtest.pkg().BGZF_EOF)
    # an explicit list with a repeated ordinal, as the one-GPU rehearsal uses it
    p = subprocess.run([cli, "-G1", "-g2"], stdin=open(src, "rb"), env=dict(env, HIPDEFLATE_DEVICES="0,0"), capture_output=True)
    assert p.returncode == 0 and p.stdout == outs["pipe", 1]
    p = subprocess.run([cli, "-d", "-g2"], input=outs["file", 3], env=env, capture_output=True)
    assert p.returncode == 0 and p.stdout == data
