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


def have_ref_cli():
    """the reference CLI built from /root/reference; on a GPU box its absence is an error (tests/conftest.py), here it
    is asserted again so that no test loses its interop half quietly"""
    ok = os.path.exists(REF)
    assert ok or not os.path.exists("/dev/kfd") or os.environ.get("HD_ALLOW_NO_REF") == "1", "oracle/_ref/cielbox_ref missing"
    return ok


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
    if have_ref_cli():
        # the REAL reference CLI decodes our file, and we decode the reference's
        p = subprocess.run([REF, "7bgzf", "-d"], input=blob, capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data
        p = subprocess.run([REF, "7bgzf", "-l1", "-@4"], input=data, capture_output=True, timeout=300)
        assert p.returncode == 0
        rc, back, err = run(["-d"], p.stdout)
        assert rc == 0 and back == data, err


@pytest.mark.parametrize("level", [1, 6])
def test_migz_framing_roundtrip(level):
    """1 MiB members (coded as flushed 0xff00-byte segments, hd_segment.hpp): plain gzip readers, our reader
    and the reference's 7migz all take them back."""
    data = bytes(hdtest.synth().text_like(3 * 1024 * 1024 + 17, seed=10))
    rc, blob, err = run(["-M", "-b1024", "-G%d" % level], data)
    assert rc == 0, err
    assert blob[:16] == bytes.fromhex("1f8b08040000000000ff08004d5a0400")
    assert gzip.decompress(blob) == data
    rc, back, err = run(["-d"], blob)
    assert rc == 0 and back == data, err
    if have_ref_cli():
        p = subprocess.run([REF, "7migz", "-d"], input=blob, capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data


DZ = os.path.join(hdtest.ROOT, "7bgzf_amd", "hd7dictzip")


@pytest.mark.parametrize("level,extreme", [(1, False), (6, False), (2, True), (0, False)])
def test_dictzip_writer_and_reader_against_the_reference(tmp_path, level, extreme):
    """hd7dictzip (applet/7dictzip.c re-shaped into batches): the file is a plain gzip member (Python's
    gzip checks CRC-32 and ISIZE, which come from folding the kernel's per-chunk CRCs), carries the 'RA'
    chunk table, every chunk is the kernel's full-flush form (== twin), the REAL reference's 7dictzip
    reads it, and we read the reference's file."""
    import struct
    assert os.path.exists(DZ)
    bs = 0xff00 if extreme else 58315
    data = bytes(hdtest.synth().fastq_like(9 * bs + 4321, seed=31)) + bytes(hdtest.synth().random_bytes(bs + 5))
    fi, fo = str(tmp_path / "in.bin"), str(tmp_path / "out.dz")
    open(fi, "wb").write(data)
    p = subprocess.run([DZ, "-G%d" % level] + (["-X"] if extreme else []) + [fi, fo], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    assert "compression level = %d (hip)" % level in p.stderr.decode() and "done." in p.stderr.decode()
    d = open(fo, "rb").read()
    assert gzip.decompress(d) == data
    nchunks = (len(data) + bs - 1) // bs
    assert d[:10] == bytes.fromhex("1f8b0804000000000003")
    xlen, ra, sublen, ver, chlen, chcnt = struct.unpack("<H2sHHHH", d[10:22])
    assert (xlen, ra, sublen, ver, chlen, chcnt) == (10 + 2 * nchunks, b"RA", 6 + 2 * nchunks, 1, bs, nchunks)
    sizes = struct.unpack("<%dH" % nchunks, d[22:22 + 2 * nchunks])
    pos = 22 + 2 * nchunks
    for i, n in enumerate(sizes):
        r, twin = hdtest.oracle_twin_flush(data[i * bs:(i + 1) * bs], level)
        assert r == 0 and d[pos:pos + n] == twin, i
        pos += n
    assert d[pos:pos + 2] == b"\x03\x00" and len(d) == pos + 10
    p = subprocess.run([DZ, "-d", fo], capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout == data, p.stderr.decode()
    # damage: a flipped bit in a chunk or in the CRC is reported
    bad = bytearray(d)
    bad[-5] ^= 1
    open(fo, "wb").write(bytes(bad))
    p = subprocess.run([DZ, "-d", fo], capture_output=True, timeout=300)
    assert p.returncode != 0 and "mismatch" in p.stderr.decode()
    if have_ref_cli():
        open(fo, "wb").write(d)
        p = subprocess.run([REF, "7dictzip", "-cd", fo], capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data
        fr = str(tmp_path / "ref.dz")
        p = subprocess.run([REF, "7dictzip", "-cl6"] + (["-X"] if extreme else []) + [fi, fr], capture_output=True, timeout=300)
        assert p.returncode == 0
        p = subprocess.run([DZ, "-d", fr], capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data, p.stderr.decode()


RZ = os.path.join(hdtest.ROOT, "7bgzf_amd", "hd7razf")


@pytest.mark.parametrize("level,nbytes", [(1, 9 * 32768 + 4321), (6, 32768), (3, 100), (0, 3 * 32768)])
def test_razf_writer_and_reader_against_the_reference(tmp_path, level, nbytes):
    """hd7razf (applet/7razf.c in batches): one gzip member (zlib checks CRC-32 / ISIZE) + big-endian index;
    every chunk but the last is the kernel's full-flush form, the last its ordinary stream (== twin); the
    index is what the reference writes for the same chunk sizes; the REAL 7razf reads it, we read its."""
    import struct
    import zlib
    assert os.path.exists(RZ)
    data = (bytes(hdtest.synth().fastq_like(nbytes, seed=41)) + bytes(hdtest.synth().random_bytes(40000)))[:nbytes]
    fi, fo = str(tmp_path / "in.bin"), str(tmp_path / "out.raz")
    open(fi, "wb").write(data)
    p = subprocess.run([RZ, "-G%d" % level, fi], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    d = p.stdout
    open(fo, "wb").write(d)
    assert d[:19] == bytes.fromhex("1f8b0804000000000003070052415a46018000")
    z = zlib.decompressobj(31)
    assert z.decompress(d) == data
    nchunks = (len(data) + 32767) // 32768
    total, index_at = struct.unpack(">QQ", d[-16:])
    assert total == len(data) and len(z.unused_data) == len(d) - index_at
    tb = struct.unpack(">I", d[index_at:index_at + 4])[0]
    assert tb == nchunks - 1 and len(d) == index_at + 4 + 8 + 4 * tb + 16
    bin0 = struct.unpack(">Q", d[index_at + 4:index_at + 12])[0]
    cells = struct.unpack(">%dI" % tb, d[index_at + 12:index_at + 12 + 4 * tb])
    pos = 19
    for k in range(nchunks):
        chunk = data[k * 32768:(k + 1) * 32768]
        r, twin = (hdtest.oracle_twin_flush if k < nchunks - 1 else hdtest.oracle_twin)(chunk, level)
        assert r == 0 and d[pos:pos + len(twin)] == twin, k
        if k >= 1:
            assert bin0 + cells[k - 1] == pos
        pos += len(twin)
    assert pos + 8 == index_at
    p = subprocess.run([RZ, "-d", fo], capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout == data, p.stderr.decode()
    bad = bytearray(d)
    bad[index_at - 6] ^= 4
    open(fo, "wb").write(bytes(bad))
    p = subprocess.run([RZ, "-d", fo], capture_output=True, timeout=300)
    assert p.returncode != 0 and "mismatch" in p.stderr.decode()
    if have_ref_cli():
        open(fo, "wb").write(d)
        p = subprocess.run([REF, "7razf", "-cd", fo], capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data
        p = subprocess.run([REF, "7razf", "-cl6", fi], capture_output=True, timeout=300)
        assert p.returncode == 0
        fr = str(tmp_path / "ref.raz")
        open(fr, "wb").write(p.stdout)
        p = subprocess.run([RZ, "-d", fr], capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data, p.stderr.decode()


GZA = os.path.join(hdtest.ROOT, "7bgzf_amd", "hd7gzinga")


@pytest.mark.parametrize("level,nbytes", [(1, 7 * 102400 + 999), (6, 102400), (2, 0), (0, 2 * 102400)])
def test_gzinga_writer_and_reader_against_the_reference(tmp_path, level, nbytes):
    """hd7gzinga (applet/7gzinga.c in batches): 100 KiB members with an empty comment + the index member;
    a plain gzip reader reads the whole file, every member's payload == twin, the index text is the
    reference's, the REAL 7gzinga reads it and we read its."""
    assert os.path.exists(GZA)
    data = (bytes(hdtest.synth().text_like(nbytes, seed=51)) + bytes(hdtest.synth().random_bytes(3000)))[:nbytes]
    fo = str(tmp_path / "out.gz")
    p = subprocess.run([GZA, "-G%d" % level], input=data, capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    d = p.stdout
    open(fo, "wb").write(d)
    assert gzip.decompress(d) == data
    pos, ends = 0, []
    for k in range((len(data) + 102399) // 102400):
        chunk = data[k * 102400:(k + 1) * 102400]
        r, twin = hdtest.oracle_twin(chunk, level)
        assert r == 0
        member = bytes.fromhex("1f8b08100000000000ff00") + twin + hdtest.oracle_crc32(chunk).to_bytes(4, "little") + \
            len(chunk).to_bytes(4, "little")
        assert d[pos:pos + len(member)] == member, k
        pos += len(member)
        ends.append(pos)
    index = bytes.fromhex("1f8b08100000000000ff") + "".join("%d:%d;" % (k, e) for k, e in enumerate(ends)).encode() + \
        bytes.fromhex("0003000000000000000000")
    assert d[pos:] == index
    p = subprocess.run([GZA, "-d", fo], capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout == data, p.stderr.decode()
    if nbytes:
        bad = bytearray(d)
        bad[ends[0] - 7] ^= 1                                        # a CRC-32 byte of the first member
        open(fo, "wb").write(bytes(bad))
        p = subprocess.run([GZA, "-d", fo], capture_output=True, timeout=300)
        assert p.returncode != 0 and "mismatch" in p.stderr.decode()
    if have_ref_cli() and nbytes:
        open(fo, "wb").write(d)
        p = subprocess.run([REF, "7gzinga", "-cd", fo], capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data
        p = subprocess.run([REF, "7gzinga", "-cl6"], input=data, capture_output=True, timeout=300)
        assert p.returncode == 0
        fr = str(tmp_path / "ref.gz")
        open(fr, "wb").write(p.stdout)
        p = subprocess.run([GZA, "-d", fr], capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data, p.stderr.decode()


CS = os.path.join(hdtest.ROOT, "7bgzf_amd", "hd7ciso")


@pytest.mark.parametrize("level,threshold", [(1, 100), (6, 100), (2, 50)])
def test_ciso_writer_and_reader_against_the_reference(tmp_path, level, threshold):
    """hd7ciso (applet/7ciso.c in batches): 2048-byte sectors, the tiny-block end of the path (several
    thousand blocks per call).  Header and offset table as the reference lays them out, every compressed
    sector == twin, sectors over the threshold stored plain with bit 31 set; the REAL 7ciso reads our
    file and we read its."""
    import struct
    assert os.path.exists(CS)
    s = hdtest.synth()
    data = bytes(s.text_like(1 << 20, seed=61)) + bytes(300000) + bytes(s.random_bytes(200000)) + \
        bytes(s.fastq_like(1 << 20, seed=62)) + b"tail of the image, not a whole sector"
    fi, fo = str(tmp_path / "in.iso"), str(tmp_path / "out.cso")
    open(fi, "wb").write(data)
    p = subprocess.run([CS, "-G%d" % level, "-t%d" % threshold, fi, fo], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    d = open(fo, "rb").read()
    nblk = (len(data) + 2047) // 2048
    assert d[:24] == b"CISO" + struct.pack("<IQIBBH", 24, len(data), 2048, 1, 0, 0)
    idx = struct.unpack("<%dI" % (nblk + 1), d[24:24 + 4 * (nblk + 1)])
    assert idx[0] & 0x7fffffff == 24 + 4 * (nblk + 1) and idx[-1] == len(d)
    plain = 0
    for k in range(nblk):
        sector = data[k * 2048:(k + 1) * 2048]
        a, b = idx[k] & 0x7fffffff, idx[k + 1] & 0x7fffffff
        r, twin = hdtest.oracle_twin(sector, level)
        assert r == 0
        if len(twin) > 2048 * threshold // 100:
            assert idx[k] >> 31 and d[a:b] == sector, k
            plain += 1
        else:
            assert not idx[k] >> 31 and d[a:b] == twin, k
    assert 0 < plain < nblk
    p = subprocess.run([CS, "-d"], input=d, capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout == data, p.stderr.decode()
    if have_ref_cli():
        p = subprocess.run([REF, "7ciso", "-cd"], input=d, capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data
        fr = str(tmp_path / "ref.cso")
        p = subprocess.run([REF, "7ciso", "-l6", "-t%d" % threshold, fi, fr], capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        p = subprocess.run([CS, "-d"], input=open(fr, "rb").read(), capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data, p.stderr.decode()


DX = os.path.join(hdtest.ROOT, "7bgzf_amd", "hd7daxcr")


@pytest.mark.parametrize("level", [1, 6, 0])
def test_daxcr_writer_and_reader_against_the_reference(tmp_path, level):
    """hd7daxcr (applet/7daxcr.c in batches): 8192-byte frames as RFC 1950 members made on the device
    (HD_FRAME_ZLIB: header, twin payload, Adler-32), 32-bit offset and 16-bit size tables as the
    reference lays them out; zlib reads every frame, the REAL 7daxcr reads our file and we read its."""
    import struct
    import zlib
    assert os.path.exists(DX)
    s = hdtest.synth()
    data = bytes(s.text_like(600000, seed=71)) + bytes(100000) + bytes(s.random_bytes(50000)) + \
        bytes(s.fastq_like(400000, seed=72)) + b"a ragged last frame"
    fi, fo = str(tmp_path / "in.iso"), str(tmp_path / "out.dax")
    open(fi, "wb").write(data)
    p = subprocess.run([DX, "-G%d" % level, fi, fo], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    d = open(fo, "rb").read()
    nblk = (len(data) + 8191) // 8192
    assert d[:32] == b"DAX\0" + struct.pack("<III", len(data), 1, 0) + bytes(16)
    idx = struct.unpack("<%dI" % nblk, d[32:32 + 4 * nblk])
    sizes = struct.unpack("<%dH" % nblk, d[32 + 4 * nblk:32 + 6 * nblk])
    pos = 32 + 6 * nblk
    for k in range(nblk):
        frame = data[k * 8192:(k + 1) * 8192]
        assert idx[k] == pos
        m = d[pos:pos + sizes[k]]
        r, twin = hdtest.oracle_twin(frame, level)
        assert r == 0 and m[:2] == b"\x78\xda" and m[2:-4] == twin, k
        assert m[-4:] == zlib.adler32(frame).to_bytes(4, "big") and zlib.decompress(m) == frame
        pos += sizes[k]
    assert pos == len(d)
    p = subprocess.run([DX, "-d"], input=d, capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout == data, p.stderr.decode()
    if have_ref_cli():
        p = subprocess.run([REF, "7daxcr", "-cd"], input=d, capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data
        fr = str(tmp_path / "ref.dax")
        p = subprocess.run([REF, "7daxcr", "-l6", fi, fr], capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        p = subprocess.run([DX, "-d"], input=open(fr, "rb").read(), capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout == data, p.stderr.decode()


def test_rejects_garbage():
    rc, out, err = run(["-d"], b"this is not a bgzf file at all, not even close")
    assert rc != 0 and "not BGZF or corrupted" in err


def test_ld_preload_hook_takes_over_an_htslib_shaped_writer(tmp_path):
    """The product's headline use (readme.md:9-14): `BGZF_METHOD=hip1 LD_PRELOAD=libhipdeflate.so samtools ...`.
    The image has no htslib, so tests/native/fakehts.c stands in for libhts.so: an exported default
    bgzf_compress() and a threaded BGZF writer in the same shared object calling it through the PLT.
    Without the preload the stand-in's stored-block members come out; with it every member must be
    OUR level-1 member (payload == CPU twin in latency mode), produced by the batching hook under 8 threads."""
    import zlib
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native")
    so, exe = str(tmp_path / "libfakehts.so"), str(tmp_path / "hts_client")
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-pthread", "-o", so, os.path.join(here, "fakehts.c")], check=True)
    subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(here, "hts_client.c"), "-L" + str(tmp_path), "-lfakehts",
                    "-Wl,-rpath," + str(tmp_path), "-pthread"], check=True)
    data = bytes(hdtest.synth().fastq_like(40 * 0xff00 + 321))[: 40 * 0xff00 + 321]
    env = dict(os.environ)
    plain = subprocess.run([exe, "8"], input=data, capture_output=True, env=env, check=True).stdout
    assert plain[18] == 1 and len(plain) > len(data)                 # the stand-in's stored blocks
    env.update(LD_PRELOAD=hdtest.pkg().LIB_PATH, BGZF_METHOD="hip1", HIPDEFLATE_BATCH_US="1000")
    p = subprocess.run([exe, "8"], input=data, capture_output=True, env=env)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    out = p.stdout
    assert out.endswith(hdtest.pkg().BGZF_EOF) and len(out) < len(data) * 0.6
    pos, i = 0, 0
    while pos < len(out) - 28:
        total = int.from_bytes(out[pos + 16:pos + 18], "little") + 1
        member = out[pos:pos + total]
        chunk = data[i * 0xff00:(i + 1) * 0xff00]
        r, twin = hdtest.codec_twin(chunk, 1, cap=65536 - 26)             # the hook codes in latency mode
        assert r == 0 and member[18:-8] == twin, i
        assert int.from_bytes(member[-8:-4], "little") == zlib.crc32(chunk) and \
            int.from_bytes(member[-4:], "little") == len(chunk), i
        pos += total
        i += 1
    assert i == 41 and pos == len(out) - 28


def _gz_member(kind, payload, crc, isize, fname=b""):
    """one gzip member whose extra field carries the member length in each of the five ways
    _read_gz_header (applet/7bgzf.c:111-129) understands"""
    import struct
    flg = 4 | (8 if fname else 0)
    name = fname + b"\0" if fname else b""
    trailer = struct.pack("<II", crc, isize)

    def build(extra):
        return bytes([0x1f, 0x8b, 8, flg, 0, 0, 0, 0, 0, 0xff]) + struct.pack("<H", len(extra)) + extra + name
    if kind == "BC":
        total = 12 + 6 + len(name) + len(payload) + 8
        extra = b"BC\x02\x00" + struct.pack("<H", total - 1)
    elif kind == "MZ":
        extra = b"MZ\x04\x00" + struct.pack("<I", len(payload))
    elif kind == "IG1":
        total = 12 + 20 + len(name) + len(payload) + 8
        extra = b"IG\x10\x00" + struct.pack("<QQ", total, isize)
    elif kind == "IG2":
        total = 12 + 8 + len(name) + len(payload) + 8
        extra = b"IG\x04\x00" + struct.pack("<I", total)
    else:                                   # jerodsanto's mgzip: u24 member length, 0x7d tag
        total = 12 + 4 + len(name) + len(payload) + 8
        extra = struct.pack("<I", total)[:3] + b"\x7d"
    return build(extra) + payload + trailer


def test_decode_prescan_takes_every_member_kind_of_read_gz_header():
    """a13: BC, MZ, IG v1, IG v2 and mgzip members (plain and with FNAME), mixed in one file, decode through
    hd7bgzf -d; the oracle's restatement of _read_gz_header is the checker of the hand-made framing."""
    import ctypes
    import zlib
    o = hdtest.oracle()
    o.hdo_read_gz_header.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_longlong)]
    s = hdtest.synth()
    blob, want = b"", b""
    kinds = ["BC", "MZ", "IG1", "IG2", "MG"]
    for k in range(40):
        kind = kinds[k % 5]
        chunk = bytes(s.fastq_like(3000 + 1500 * k, seed=100 + k)) if k % 3 else bytes(s.text_like(40000, seed=k))
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = c.compress(chunk) + c.flush()
        m = _gz_member(kind, payload, zlib.crc32(chunk), len(chunk), fname=b"chunk%03d.txt" % k if k % 2 else b"")
        eo, el, bl = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
        n = o.hdo_read_gz_header(m, min(len(m), 128), ctypes.byref(eo), ctypes.byref(el), ctypes.byref(bl))
        assert n == len(m) - len(payload) - 8 and bl.value == len(m), (kind, n, bl.value, len(m))
        blob += m
        want += chunk
    rc, back, err = run(["-d"], blob)
    assert rc == 0 and back == want, err
    assert gzip.decompress(blob) == want


@pytest.mark.parametrize("bad", ["bsize_small", "bsize_tiny", "no_extra", "cut"])
def test_decode_prescan_rejects_corrupt_headers_like_the_reference(bad):
    """a BC member whose BSIZE + 1 is smaller than header + trailer must end as the reference's
    "not BGZF or corrupted" (-1), never as a read in front of the buffer"""
    import struct
    pkg = hdtest.pkg()
    data = bytes(hdtest.synth().fastq_like(3 * 0xff00, seed=4))
    blob = bytearray(pkg.bgzf_compress_bytes(data, 1))
    if bad == "bsize_small":
        blob[16:18] = struct.pack("<H", 20)          # total 21: below header (18) + trailer (8)
    elif bad == "bsize_tiny":
        blob[16:18] = struct.pack("<H", 1)
    elif bad == "no_extra":
        blob[3] = 0
    else:
        blob = blob[: len(blob) // 2]
    rc, back, err = run(["-d"], bytes(blob))
    if bad == "cut":
        assert rc != 0, err              # (the reference inflates the short read and fails with "inflate N")
    else:
        assert rc == 255 and "not BGZF or corrupted" in err, (rc, err)


def test_gzi_index_from_the_device_scan_and_random_access(tmp_path):
    """hd7bgzf --index: bgzip's .gzi (u64 count, then (compressed, uncompressed) offset pairs for every block start
    but the first) straight from the device's size prefix scan.  Checked against a table re-derived on the host by
    walking BSIZE; then 100 sampled members are inflated through hipdeflate_batch_inflate using ONLY the index
    (member = [coffset_i, coffset_{i+1}), payload behind the 18-byte header) and BAM virtual offsets are formed."""
    import struct
    import numpy as np
    pkg = hdtest.pkg()
    data = bytes(hdtest.synth().fastq_like(1500 * 0xff00 + 4321, seed=12))          # 3 pipe batches
    idx_path = str(tmp_path / "out.bgz.gzi")
    rc, blob, err = run(["-G1", "--index", idx_path], data)
    assert rc == 0, err
    raw = open(idx_path, "rb").read()
    (cnt,) = struct.unpack_from("<Q", raw)
    pairs = np.frombuffer(raw, dtype="<u8", offset=8).reshape(-1, 2)
    assert cnt == len(pairs) == 1500                                            # 1501 data blocks, the first has no record
    # the host's walk over BSIZE
    want, pos, k = [], 0, 0
    while pos < len(blob) - 28:
        if k:
            want.append((pos, k * 0xff00))
        pos += int.from_bytes(blob[pos + 16:pos + 18], "little") + 1
        k += 1
    assert pos == len(blob) - 28 and k == 1501
    assert [tuple(int(x) for x in p) for p in pairs] == want
    # random access with nothing but the index
    coff = [0] + [int(p[0]) for p in pairs] + [len(blob) - 28]
    uoff = [0] + [int(p[1]) for p in pairs] + [len(data)]
    rng = np.random.default_rng(3)
    pick = sorted(set(int(i) for i in rng.integers(0, 1501, 100)) | {0, 1500})
    streams = [blob[coff[i] + 18:coff[i + 1]] for i in pick]                   # payload + trailer (applet/7bgzf.c:328)
    caps = [uoff[i + 1] - uoff[i] for i in pick]
    outs, crc, st = pkg.batch_inflate(streams, caps)
    for j, i in enumerate(pick):
        assert st[j] == 0 and outs[j] == data[uoff[i]:uoff[i + 1]], i
        assert int(crc[j]) == int.from_bytes(blob[coff[i + 1] - 8:coff[i + 1] - 4], "little")
    # virtual offset of byte 1000 of block 7: (coffset << 16) | uoffset, as htslib's bgzf_tell
    v = (coff[7] << 16) | 1000
    assert v >> 16 == want[6][0] and v & 0xffff == 1000


def test_file_to_file_parallel_io_path(tmp_path):
    """hd7bgzf -i IN -o OUT -@N: worker threads pread() into the pinned batches and pwrite() the finished runs at the
    scanned offsets; the file is byte-identical to the stdin/stdout filter's, with the same .gzi."""
    data = bytes(hdtest.synth().fastq_like(700 * 0xff00 + 99, seed=21))
    fi, fo, fx = str(tmp_path / "in.bin"), str(tmp_path / "out.bgz"), str(tmp_path / "out.gzi")
    open(fi, "wb").write(data)
    p = subprocess.run([EXE, "-G1", "-@6", "-i", fi, "-o", fo, "--index", fx], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    rc, blob, err = run(["-G1", "--index", str(tmp_path / "ref.gzi")], data)
    assert rc == 0 and open(fo, "rb").read() == blob
    assert open(fx, "rb").read() == open(str(tmp_path / "ref.gzi"), "rb").read()
    assert gzip.decompress(blob) == data


def _png(width, height, depth, color, interlace, raw, extra_chunks=(), idat_split=3, level=6):
    import struct
    import zlib as _z

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", _z.crc32(t + d))
    z = _z.compress(raw, level)
    cut = [len(z) * k // idat_split for k in range(idat_split + 1)]
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, color, 0, 0, interlace))
    for t, d in extra_chunks:
        out += chunk(t, d)
    for k in range(idat_split):
        out += chunk(b"IDAT", z[cut[k]:cut[k + 1]])
    return out + chunk(b"tEXt", b"Comment\0made for the test") + chunk(b"IEND", b"")


def _png_chunks(blob):
    import struct
    import zlib as _z
    assert blob[:8] == b"\x89PNG\r\n\x1a\n"
    pos, out = 8, []
    while pos < len(blob):
        (n,) = struct.unpack(">I", blob[pos:pos + 4])
        t, d = blob[pos + 4:pos + 8], blob[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", blob[pos + 8 + n:pos + 12 + n])[0] == _z.crc32(t + d), t
        out.append((t, d))
        pos += 12 + n
    return out


def test_png_idat_recoder_batches_images(tmp_path):
    """hd7png (applet/7png.c re-shaped): the IDATs of several images -- RGB, paletted with PLTE/tRNS, 16-bit RGBA,
    Adam7-interlaced, one large enough for flushed segments -- are inflated in ONE batch and coded again in ONE batch
    as RFC 1950 members made on the device (78 da, raw DEFLATE == CPU twin, Adler-32); every output is a valid PNG
    (chunk CRCs), its single IDAT inflates to the same pixels, the other chunks survive (or go with -t)."""
    import zlib
    import numpy as np
    rng = np.random.default_rng(5)
    PNG = os.path.join(hdtest.ROOT, "7bgzf_amd", "hd7png")

    def rows(h, rowbytes, smooth=True):
        a = np.cumsum(rng.integers(-2, 3, (h, rowbytes)), axis=1).astype(np.uint8) if smooth else rng.integers(0, 256, (h, rowbytes), dtype=np.uint8)
        return b"".join(b"\0" + bytes(r) for r in a)
    adam = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    w7, h7 = 37, 29
    raw7 = b"".join(rows(-(-(h7 - y0) // dy), -(-(w7 - x0) // dx) * 3) for x0, y0, dx, dy in adam if w7 > x0 and h7 > y0)
    images = [
        (640, 480, 8, 2, 0, rows(480, 640 * 3), ()),
        (100, 60, 8, 3, 0, rows(60, 100), ((b"PLTE", bytes(range(256)) * 3), (b"tRNS", bytes(256)))),
        (200, 90, 16, 6, 0, rows(90, 200 * 8), ((b"gAMA", b"\0\0\xb1\x8f"),)),
        (w7, h7, 8, 2, 1, raw7, ()),
        (1400, 900, 8, 2, 0, rows(900, 1400 * 3), ()),             # 3.8 MB of pixels: coded in flushed segments
        (5, 3, 1, 0, 0, b"\0\xa8" * 3, ()),
    ]
    args, want = [], []
    for k, (w, h, depth, color, il, raw, extra) in enumerate(images):
        fi, fo = str(tmp_path / ("in%d.png" % k)), str(tmp_path / ("out%d.png" % k))
        open(fi, "wb").write(_png(w, h, depth, color, il, raw, extra, idat_split=1 + k % 3))
        args += [fi, fo]
        want.append(raw)
    for level, strip in ((1, False), (6, True)):
        p = subprocess.run([PNG, "-G%d" % level] + (["-t"] if strip else []) + args, capture_output=True, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-500:]
        err = p.stderr.decode()
        assert err.count("recompressed length=") == len(images) and "Done." in err
        for k, raw in enumerate(want):
            src, dst = _png_chunks(open(args[2 * k], "rb").read()), _png_chunks(open(args[2 * k + 1], "rb").read())
            idat = [d for t, d in dst if t == b"IDAT"]
            assert len(idat) == 1 and idat[0][:2] == b"\x78\xda" and zlib.decompress(idat[0]) == raw, k
            r, twin = hdtest.oracle_twin(raw, level, cap=len(raw) + len(raw) // 2 + 4096)
            assert r == 0 and idat[0][2:-4] == twin, k                                # the device's member == CPU twin
            keep = [(t, d) for t, d in src if t != b"IDAT" and (not strip or t in (b"IHDR", b"PLTE", b"tRNS", b"IEND"))]
            assert [(t, d) for t, d in dst if t != b"IDAT"] == keep, k
            assert dst[-1][0] == b"IEND"
    # the filter form of the reference, and a file that is not a PNG
    p = subprocess.run([PNG, "-G1"], input=open(args[0], "rb").read(), capture_output=True, timeout=300)
    assert p.returncode == 0 and zlib.decompress(b"".join(d for t, d in _png_chunks(p.stdout) if t == b"IDAT")) == want[0]
    p = subprocess.run([PNG, "-G1"], input=b"GIF89a not a png at all", capture_output=True, timeout=300)
    assert p.returncode != 0 and b"not PNG file" in p.stderr
