"""The reference ITSELF on the hip backend (VERDICT r3 "what's missing" 4 / "next" 1c).

oracle/_ref/cielbox_hip is the reference's multi-call CLI built by oracle/Makefile from /root/reference with
integration/7bgzf-hip.patch applied (DEFLATE_HIP + -G/--hip in applet/7bgzf.c and applet/7migz.c, hip_inflate behind
zlibutil_auto_inflate) and linked against 7bgzf_amd/libhipdeflate.so.  Nothing of ours is between the reference's loops
and the codecs: applet/7bgzf.c:159-277 creates a thread per block whose start routine is zlibutil_buffer_code ->
hip_deflate; :306-360 a thread per member -> zlibutil_auto_inflate -> hip_inflate.  Checked against the unpatched
reference (cielbox_ref) in both directions.  GPU box only; both binaries travel with the tree (git-ignored test
infrastructure, like libref.so)."""
import gzip
import os
import subprocess

import pytest

import hdtest

pytestmark = pytest.mark.gpu
HIP = os.path.join(hdtest.ROOT, "oracle", "_ref", "cielbox_hip")
REF = os.path.join(hdtest.ROOT, "oracle", "_ref", "cielbox_ref")


def need():
    ok = os.path.exists(HIP) and os.path.exists(REF)
    assert ok or os.environ.get("HD_ALLOW_NO_REF") == "1", "oracle/_ref/cielbox_hip / cielbox_ref missing: build where /root/reference exists"
    if not ok:
        pytest.skip("no reference build on this box (HD_ALLOW_NO_REF=1)")


def run(exe, args, data, env=None):
    p = subprocess.run([exe] + args, input=data, capture_output=True, timeout=600, env=env)
    return p.returncode, p.stdout, p.stderr.decode(errors="replace")


@pytest.mark.parametrize("level,threads", [(1, 16), (6, 16), (2, 1)])
def test_reference_7bgzf_encodes_through_hip_deflate(level, threads):
    """`cielbox_hip 7bgzf -G<level> -@16`: the reference's thread-per-block loop on hip_deflate.  Its file is BGZF that
    gzip, the unpatched reference and hd7bgzf all read; every member's payload is what the per-block codec gives for
    that block (the latency form's twin), i.e. the reference framed our bytes untouched."""
    need()
    pkg = hdtest.pkg()
    data = bytes(hdtest.synth().fastq_like(40 * 0xff00 + 1234, seed=31))
    rc, blob, err = run(HIP, ["7bgzf", "-G%d" % level, "-@%d" % threads], data)
    assert rc == 0, err
    assert "compression level = %d (hip)" % level in err
    assert blob.endswith(pkg.BGZF_EOF)
    assert gzip.decompress(blob) == data
    rc, back, err = run(REF, ["7bgzf", "-d", "-@4"], blob)
    assert rc == 0 and back == data, err
    assert pkg.bgzf_decompress_bytes(blob) == data
    # payload of every member == hip_deflate of its block (threads > 1: 0xff00-byte blocks, applet/7bgzf.c:146-147)
    if threads > 1:
        pos = 0
        for (o, ln, isz) in pkg.bgzf_scan(blob)[:-1]:
            r, z = pkg.hip_deflate(data[pos:pos + isz], level, cap=65536 - 18 - 8)
            assert r == 0 and blob[o:o + ln - 8] == z, pos
            pos += isz
        assert pos == len(data)


def test_reference_7bgzf_decodes_through_hip_inflate():
    """`cielbox_hip 7bgzf -d -@16`: a thread per member -> zlibutil_auto_inflate -> hip_inflate (USE_HIP_INFLATE), on files
    written by the unpatched reference (libdeflate 1 / 6, zlib 6, slz: multi-block members, dynamic, static and stored
    blocks) and by hd7bgzf; and a damaged member makes it fail as the reference does."""
    need()
    s = hdtest.synth()
    data = bytes(s.fastq_like(24 * 0xff00 + 77, seed=32)) + bytes(s.text_like(300000, seed=33)) + os.urandom(70000)
    for args in (["-l1", "-@4"], ["-l6", "-@4"], ["-z6", "-@4"], ["-s1", "-@4"], ["-l1"]):
        rc, blob, err = run(REF, ["7bgzf"] + args, data)
        assert rc == 0, err
        rc, back, err = run(HIP, ["7bgzf", "-d", "-@16"], blob)
        assert rc == 0 and back == data, (args, err[-500:])
    exe = os.path.join(hdtest.ROOT, "7bgzf_amd", "hd7bgzf")
    rc, blob, err = run(exe, ["-G6"], data)
    assert rc == 0
    rc, back, err = run(HIP, ["7bgzf", "-d", "-@16"], blob)
    assert rc == 0 and back == data
    bad = bytearray(blob)
    bad[18 + 40] ^= 0x10                                               # inside the first member's payload
    rc_h, out_h, _ = run(HIP, ["7bgzf", "-d", "-@4"], bytes(bad))
    rc_r, out_r, _ = run(REF, ["7bgzf", "-d", "-@4"], bytes(bad))
    assert (rc_h != 0) == (rc_r != 0) or out_h == out_r                # same verdict, or (both accept) the same bytes


def test_reference_7migz_both_ways_on_the_hip_backend():
    """`cielbox_hip 7migz -G6 -b1024` (config 5's container through the reference's own loop) and `7migz -d` on it and on
    the unpatched reference's file."""
    need()
    data = bytes(hdtest.synth().text_like(3 * (1 << 20) + 4242, seed=34))
    rc, blob, err = run(HIP, ["7migz", "-G6", "-b1024", "-@3"], data)
    assert rc == 0, err
    assert blob[:16] == bytes.fromhex("1f8b08040000000000ff08004d5a0400")
    assert gzip.decompress(blob) == data
    rc, back, err = run(REF, ["7migz", "-d"], blob)
    assert rc == 0 and back == data, err
    rc, back, err = run(HIP, ["7migz", "-d", "-@3"], blob)
    assert rc == 0 and back == data, err
    rc, blob_ref, err = run(REF, ["7migz", "-l6", "-b1024"], data)
    assert rc == 0
    rc, back, err = run(HIP, ["7migz", "-d", "-@3"], blob_ref)
    assert rc == 0 and back == data, err


def test_reference_on_two_device_contexts():
    """the same CLI with HIPDEFLATE_DEVICES=0,0: the per-block codecs spread their contexts over both entries"""
    need()
    data = bytes(hdtest.synth().fastq_like(64 * 0xff00, seed=35))
    env = dict(os.environ, HIPDEFLATE_DEVICES="0,0")
    rc, blob, err = run(HIP, ["7bgzf", "-G1", "-@16"], data, env)
    assert rc == 0, err
    rc, blob1, err = run(HIP, ["7bgzf", "-G1", "-@16"], data)
    assert rc == 0 and blob == blob1
    rc, back, err = run(HIP, ["7bgzf", "-d", "-@16"], blob, env)
    assert rc == 0 and back == data, err
