"""hd7bgzf (7bgzf_amd/csrc/hd_bgzf_host.c): the batch-loop replacement of
applet/7bgzf.c's per-block loop, as a stdin->stdout filter.  GPU box only."""
import gzip
import os
import subprocess

import pytest

import hdtest

pytestmark = pytest.mark.gpu
EXE = os.path.join(hdtest.ROOT, "7bgzf_amd", "hd7bgzf")
REF = os.path.join(hdtest.ROOT, "oracle", "_ref", "cielbox_ref")


def run(args, data):
    p = subprocess.run([EXE] + args, input=data, capture_output=True, timeout=300)
    return p.returncode, p.stdout, p.stderr.decode()


@pytest.mark.parametrize("level", [0, 1, 6])
def test_filter_matches_library_and_reference_format(level):
    pkg = hdtest.pkg()
    assert os.path.exists(EXE)
    data = bytes(hdtest.synth().fastq_like(5 * 1024 * 1024 + 333, seed=9))
    rc, blob, err = run(["-G%d" % level], data)
    assert rc == 0, err
    assert "compression level = %d (hip)" % level in err and "done." in err and "ellapsed time" in err
    assert blob == pkg.bgzf_compress_bytes(data, level)            # same members as the library path
    assert gzip.decompress(blob) == data
    rc, back, err = run(["-d"], blob)
    assert rc == 0 and back == data, err
    if os.path.exists(REF):
        # the REAL reference CLI decodes our file, and we decode the reference's
        p = subprocess.run([REF, "7bgzf", "-d"], input=blob, capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data
        p = subprocess.run([REF, "7bgzf", "-l1", "-@4"], input=data, capture_output=True, timeout=300)
        assert p.returncode == 0
        rc, back, err = run(["-d"], p.stdout)
        assert rc == 0 and back == data, err


def test_migz_framing_roundtrip():
    data = bytes(hdtest.synth().text_like(3 * 1024 * 1024 + 17, seed=10))
    rc, blob, err = run(["-M", "-b1024", "-G1"], data)
    assert rc == 0, err
    assert blob[:16] == bytes.fromhex("1f8b08040000000000ff08004d5a0400")
    assert gzip.decompress(blob) == data
    rc, back, err = run(["-d"], blob)
    assert rc == 0 and back == data, err
    if os.path.exists(REF):
        p = subprocess.run([REF, "7migz", "-d"], input=blob, capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data


def test_rejects_garbage():
    rc, out, err = run(["-d"], b"this is not a bgzf file at all, not even close")
    assert rc != 0 and "not BGZF or corrupted" in err
