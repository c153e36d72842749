#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (oracle/_ref/libref.so,
built by oracle/Makefile from /root/reference).  Run in the build container:

    make -C oracle ref && python tests/golden/make_golden.py

The outputs are DATA (inputs, reference outputs, verdicts), not source:

* inflate_std_vects.json -- the 151 malformed DEFLATE streams the reference's
  own test-suite holds (byte arrays of lib/isa-l/igzip/inflate_std_vects.h:4-803
  and the expected ISAL_* code of :810-962), plus the verdict of each of the
  three in-tree inflaters run here (libdeflate_inflate lib/zlibutil.c:194,
  igzip_inflate lib/zlibutil_igzip.c:93, raw zlib inflate).
* ref_streams.json -- raw-DEFLATE streams produced by every reference encoder
  on small seeded inputs (libdeflate 0/1/6/9/12, zlib 1/6/9, slz 1, miniz 1,
  store), with the SHA-256 of what they inflate to.  They cover stored, static,
  dynamic, multi-block and 15-bit-code streams.
* mutants.json -- bit-flip / truncation mutants of valid streams with the
  verdict (and output hash on success) of libdeflate_inflate: the accept/reject
  contract of the inflate path.
* full_flush.json -- chunks of the reference's own 7dictzip (applet/7dictzip.c)
  next to the raw stream the same encoder gives for the same input: pins the
  full-flush form (zlibutil_buffer_full_flush, applet/7dictzip.c:93-126) that
  HD_FRAME_RAW_FLUSH produces on the GPU.
* ratio_ref.json -- compressed sizes of the reference's encoders (libdeflate 1/2/6/9, zlib 6, slz) on seeded
  FASTQ-like / text block sets: the ratio envelope of the encoder levels.
* boundary.json -- known answers at the drop-in boundary: bgzf_compress
  (bgzf_compress.c:39) return codes/sizes/header bytes, the EOF member,
  zlibutil_buffer_code's RFC1950/1952 wrappers (lib/zlibutil.c:374-405),
  store_deflate, crc32/adler32 values.
"""
import base64
import ctypes
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hdtest  # noqa: E402

REF_TREE = os.environ.get("HD_REFERENCE", "/root/reference")


def b64(b):
    return base64.b64encode(bytes(b)).decode()


def zlib_raw_inflate(ref, data, cap):
    """raw inflate(Z_FINISH) with the reference's zlib 1.3.1; returns (ok, out)."""

    class ZS(ctypes.Structure):
        _fields_ = [("next_in", ctypes.c_void_p), ("avail_in", ctypes.c_uint), ("total_in", ctypes.c_ulong),
                    ("next_out", ctypes.c_void_p), ("avail_out", ctypes.c_uint), ("total_out", ctypes.c_ulong),
                    ("msg", ctypes.c_char_p), ("state", ctypes.c_void_p), ("zalloc", ctypes.c_void_p),
                    ("zfree", ctypes.c_void_p), ("opaque", ctypes.c_void_p), ("data_type", ctypes.c_int),
                    ("adler", ctypes.c_ulong), ("reserved", ctypes.c_ulong)]

    src = hdtest.as_u8(data)
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    z = ZS()
    ref.inflateInit2_.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    assert ref.inflateInit2_(ctypes.byref(z), -15, b"1.3.1", ctypes.sizeof(ZS)) == 0
    z.next_in = src.ctypes.data
    z.avail_in = len(src)
    z.next_out = dst.ctypes.data
    z.avail_out = cap
    r = ref.inflate(ctypes.byref(z), 4)  # Z_FINISH
    out = bytes(dst[: z.total_out])
    ref.inflateEnd(ctypes.byref(z))
    return r == 1, out  # Z_STREAM_END


ISAL_CODES = {"ISAL_DECOMP_OK": 0, "ISAL_END_INPUT": 1, "ISAL_OUT_OVERFLOW": 2, "ISAL_INVALID_BLOCK": -1,
              "ISAL_INVALID_SYMBOL": -2, "ISAL_INVALID_LOOKBACK": -3}


def isal_stateless(ref, data, cap):
    """isal_inflate_stateless (lib/isa-l/igzip/igzip_inflate.c:2146) called directly, the
    way igzip_rand_test.c:2538 drives it; struct inflate_state head fields per
    lib/isa-l/include/igzip_lib.h:511-518."""
    import struct
    src = hdtest.as_u8(data)
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    st = (ctypes.c_uint8 * (1 << 18))()
    ref.isal_inflate_init(st)
    struct.pack_into("<QII", st, 0, dst.ctypes.data, cap, 0)
    struct.pack_into("<Q", st, 16, src.ctypes.data)
    struct.pack_into("<I", st, 32, len(src))
    return ref.isal_inflate_stateless(st)


def gen_std_vects(ref):
    text = open(os.path.join(REF_TREE, "lib/isa-l/igzip/inflate_std_vects.h")).read()
    arrays = {}
    for m in re.finditer(r"uint8_t\s+(std_vect_\d+)\[\]\s*=\s*\{([^}]*)\}", text):
        arrays[m.group(1)] = bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", m.group(2)))
    table = re.findall(r"\{\s*(std_vect_\d+),\s*sizeof\(\1\),\s*(ISAL_\w+)\s*\}", text)
    assert len(table) == 151 and len(arrays) == 151
    out = []
    for name, err in table:
        v = arrays[name]
        cap = 1 << 20
        r_ld, _ = hdtest.call_dec(ref.libdeflate_inflate, v, cap)
        r_ig, _ = hdtest.call_dec(ref.igzip_inflate, v, cap)
        ok_z, _ = zlib_raw_inflate(ref, v, cap)
        r_isal = isal_stateless(ref, v, cap)
        assert r_isal == ISAL_CODES[err], (name, r_isal, err)
        out.append({"name": name, "data": b64(v), "isal_expected": err, "isal_stateless": r_isal,
                    "libdeflate": r_ld, "igzip_adapter": r_ig, "zlib_stream_end": ok_z})
    json.dump(out, open(os.path.join(HERE, "inflate_std_vects.json"), "w"), indent=0)
    print("inflate_std_vects.json:", len(out), "vectors; rejected by libdeflate:",
          sum(1 for o in out if o["libdeflate"] != 0))


ENCODERS = [("libdeflate", 0), ("libdeflate", 1), ("libdeflate", 6), ("libdeflate", 9), ("libdeflate", 12),
            ("zlib", 1), ("zlib", 6), ("zlib", 9), ("slz", 1), ("miniz", 1), ("store", 0)]


def ref_encode(ref, name, level, data):
    if name == "slz":
        ref.slz_initialize()
    f = getattr(ref, name + "_deflate")
    r, z = hdtest.call_enc(f, data, level, cap=len(data) * 2 + 4096)
    assert r == 0, (name, level, r)
    return z


def gen_ref_streams(ref):
    corpus = hdtest.corpus_small()
    # keep the committed file small: each 64 KiB input only through the encoders
    # that give it a distinct block structure (SURVEY.md section 4 "stream mix")
    big = {"fastq_ff00": {("libdeflate", 1), ("libdeflate", 6), ("zlib", 6), ("slz", 1)},
           "text_ff00": {("libdeflate", 6)}, "mixed": {("libdeflate", 12), ("zlib", 1)},
           "zeros_64k": {("libdeflate", 1), ("zlib", 6), ("slz", 1)}}
    out = []
    for cname, data in corpus.items():
        if cname == "empty":
            encs = [e for e in ENCODERS if e[0] != "store"]  # store_deflate emits nothing for n == 0
        else:
            encs = ENCODERS
        for name, level in encs:
            if len(data) > 30000 and (name, level) not in big.get(cname, ()):
                continue
            z = ref_encode(ref, name, level, data)
            r, o = hdtest.call_dec(ref.libdeflate_inflate, z, len(data))
            assert r == 0 and o == data
            out.append({"input": cname, "encoder": name, "level": level, "stream": b64(z),
                        "out_len": len(data), "out_sha256": hdtest.sha(data)})
    json.dump(out, open(os.path.join(HERE, "ref_streams.json"), "w"), indent=0)
    print("ref_streams.json:", len(out), "streams,", sum(len(o["stream"]) for o in out) * 3 // 4, "bytes")


FULL_INPUTS = [("fastq", 101), ("fastq", 102), ("text", 103), ("text", 104), ("mixed", 105)]
FULL_ENCODERS = [("libdeflate", 1), ("libdeflate", 6), ("libdeflate", 9), ("libdeflate", 12), ("zlib", 1), ("zlib", 6),
                 ("zlib", 9), ("slz", 1), ("miniz", 1)]


def full_input(kind, seed):
    """the 0xff00-byte inputs of ref_streams_full.json: regenerated from (kind, seed) by the tests, not stored"""
    s = hdtest.synth()
    if kind == "fastq":
        return bytes(s.fastq_like(0xff00, seed=seed))
    if kind == "text":
        return bytes(s.text_like(0xff00, seed=seed))
    # mixed: text, a far repeat of it (32 KiB-class distances), DNA-like, noise, zeros
    t = bytes(s.text_like(20000, seed=seed))
    return (t + bytes(s.fastq_like(14000, seed=seed + 1)) + t[:12000] + bytes(s.random_bytes(6000, seed=seed + 2)) +
            bytes(3000) + t[5000:])[:0xff00].ljust(0xff00, b"x")


def gen_ref_streams_full(ref):
    """BASELINE config 3's shape: FULL 0xff00-byte blocks through every reference encoder that the decode path
    will meet (32 KiB-distance copies, 15-bit codes, multi-block members from the 8,192-sequence cap).  Only the
    STREAM and the SHA-256 of its inflation are stored; plus one 1 MiB MiGz member made by the real `7migz -l6
    -b1024` (oracle/_ref/cielbox_ref)."""
    import subprocess
    import tempfile
    out = []
    for kind, seed in FULL_INPUTS:
        data = full_input(kind, seed)
        for name, level in FULL_ENCODERS:
            z = ref_encode(ref, name, level, data)
            r, o = hdtest.call_dec(ref.libdeflate_inflate, z, len(data))
            assert r == 0 and o == data
            out.append({"kind": kind, "seed": seed, "encoder": name, "level": level, "stream": b64(z),
                        "out_len": len(data), "out_sha256": hdtest.sha(data)})
    box = os.path.join(ROOT, "oracle", "_ref", "cielbox_ref")
    text = bytes(hdtest.synth().text_like(1 << 20, seed=106))
    p = subprocess.run([box, "7migz", "-l6", "-b1024"], input=text, capture_output=True, check=True)
    m = p.stdout
    assert m[:4] == b"\x1f\x8b\x08\x04" and m[12:16] == b"MZ\x04\x00"
    paylen = int.from_bytes(m[16:20], "little")
    assert 20 + paylen + 8 == len(m)
    out.append({"kind": "migz_text_1mib", "seed": 106, "encoder": "7migz(libdeflate)", "level": 6, "stream": b64(m[20:20 + paylen]),
                "member_header": m[:20].hex(), "member_trailer": m[-8:].hex(), "out_len": len(text), "out_sha256": hdtest.sha(text)})
    json.dump(out, open(os.path.join(HERE, "ref_streams_full.json"), "w"), indent=0)
    print("ref_streams_full.json:", len(out), "streams,", sum(len(o["stream"]) for o in out) * 3 // 4, "bytes")


def gen_mutants(ref):
    rng = np.random.default_rng(2025)
    s = hdtest.synth()
    bases = []
    for k, data in (("fastq", bytes(s.fastq_like(1500, seed=7))), ("text", bytes(s.text_like(1200, seed=8))),
                    ("rep", b"abcdefgh" * 100 + bytes(200))):
        for name, level in (("libdeflate", 1), ("libdeflate", 6), ("libdeflate", 12), ("zlib", 6), ("slz", 1)):
            bases.append((k, name, level, data, ref_encode(ref, name, level, data)))
    out = []
    for k, name, level, data, z in bases:
        for _ in range(40):
            m = bytearray(z)
            kind = int(rng.integers(0, 4))
            if kind < 3:
                for _ in range(kind + 1):
                    bit = int(rng.integers(0, len(m) * 8))
                    m[bit >> 3] ^= 1 << (bit & 7)
            else:
                m = m[: max(1, len(m) - int(rng.integers(1, 16)))]
            cap = len(data) + 64
            r, o = hdtest.call_dec(ref.libdeflate_inflate, bytes(m), cap)
            r_ig, o_ig = hdtest.call_dec(ref.igzip_inflate, bytes(m), cap)
            okz, oz = zlib_raw_inflate(ref, bytes(m), cap)
            out.append({"base": "%s/%s%d" % (k, name, level), "stream": b64(m), "cap": cap, "libdeflate": r,
                        "out_sha256": hdtest.sha(o) if r == 0 else None, "out_len": len(o) if r == 0 else None,
                        "igzip_adapter": r_ig, "zlib_stream_end": okz})
    json.dump(out, open(os.path.join(HERE, "mutants.json"), "w"), indent=0)
    acc = sum(1 for o in out if o["libdeflate"] == 0)
    print("mutants.json:", len(out), "mutants; accepted by libdeflate:", acc)


def gen_boundary(ref):
    s = hdtest.synth()
    fq = bytes(s.fastq_like(0xff00))
    res = {}
    os.environ["BGZF_METHOD"] = "libdeflate1"
    ref.bgzf_compress.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p,
                                  ctypes.c_size_t, ctypes.c_int]

    def hook(data, cap):
        src = hdtest.as_u8(data)
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        n = ctypes.c_size_t(cap)
        r = ref.bgzf_compress(dst.ctypes.data, ctypes.byref(n), src.ctypes.data if len(src) else None, len(src), -1)
        return r, n.value, bytes(dst[: n.value]) if r == 0 else b""

    r, n, m = hook(fq, 0x10000)
    payload = m[18:-8]
    r2, inflated = hdtest.call_dec(ref.libdeflate_inflate, payload, len(fq))
    assert r == 0 and r2 == 0 and inflated == fq
    res["hook_fastq_ff00"] = {"ret": r, "dlen": n, "header18": m[:18].hex(), "trailer8": m[-8:].hex(),
                              "bsize_plus1": int.from_bytes(m[16:18], "little") + 1,
                              "first_block_bfinal": payload[0] & 1, "first_block_btype": (payload[0] >> 1) & 3,
                              "input_sha256": hdtest.sha(fq)}
    r, n, m = hook(b"", 0x10000)
    res["hook_eof"] = {"ret": r, "dlen": n, "member": m.hex()}
    r, n, _ = hook(b"", 27)
    res["hook_eof_cap27"] = {"ret": r}
    r, n, _ = hook(fq[:100], 20)
    res["hook_cap20"] = {"ret": r}
    r, n, _ = hook(fq[:100], 25)
    res["hook_cap25"] = {"ret": r}

    # checksums
    res["crc32"] = {k: ref.crc32(0, v, len(v)) & 0xffffffff for k, v in
                    (("empty", b""), ("a", b"a"), ("123456789", b"123456789"), ("fastq_ff00", fq))}
    ref.crc32_gzip_refl.restype = ctypes.c_uint32
    res["crc32_gzip_refl"] = {k: ref.crc32_gzip_refl(0, v, ctypes.c_uint64(len(v))) for k, v in
                              (("123456789", b"123456789"), ("fastq_ff00", fq))}
    res["adler32"] = {k: ref.adler32(1, v, len(v)) & 0xffffffff for k, v in
                      (("empty", b""), ("123456789", b"123456789"), ("fastq_ff00", fq))}

    # store_deflate known answers (lib/zlibutil.c:302)
    st = {}
    for k, v in (("abc", b"abc"), ("n65535", bytes(65535)), ("n65536", bytes(65536)), ("n70000", fq + fq[:4720])):
        r, z = hdtest.call_enc(ref.store_deflate, v, 0, cap=len(v) + 100)
        st[k] = {"ret": r, "len": len(z), "sha256": hdtest.sha(z), "head5": z[:5].hex()}
    r, z = hdtest.call_enc(ref.store_deflate, b"abc", 0, cap=7)
    st["abc_cap7"] = {"ret": r}
    res["store_deflate"] = st

    # zlibutil_buffer_code wrappers around store_deflate (deterministic payload)
    class ZB(ctypes.Structure):
        _fields_ = [("dest", ctypes.c_void_p), ("destLen", ctypes.c_size_t), ("source", ctypes.c_void_p),
                    ("sourceLen", ctypes.c_size_t), ("func", ctypes.c_void_p), ("encode", ctypes.c_int),
                    ("level", ctypes.c_int), ("rfc1950", ctypes.c_int), ("rfc1952", ctypes.c_int),
                    ("ret", ctypes.c_int)]

    ref.zlibutil_buffer_allocate.restype = ctypes.POINTER(ZB)
    ref.zlibutil_buffer_allocate.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
    ref.zlibutil_buffer_code.restype = ctypes.POINTER(ZB)
    ref.zlibutil_buffer_code.argtypes = [ctypes.POINTER(ZB)]
    ref.zlibutil_buffer_free.argtypes = [ctypes.POINTER(ZB)]
    wr = {}
    data = b"hello hello hello wrapper"
    for mode in ("rfc1950", "rfc1952"):
        zb = ref.zlibutil_buffer_allocate(200, len(data))
        ctypes.memmove(zb.contents.source, data, len(data))
        zb.contents.func = ctypes.cast(ref.store_deflate, ctypes.c_void_p)
        zb.contents.encode = 1
        zb.contents.level = 0
        setattr(zb.contents, mode, 1)
        ref.zlibutil_buffer_code(zb)
        out = ctypes.string_at(zb.contents.dest, zb.contents.destLen)
        if mode == "rfc1952":
            out = out[:4] + b"\0\0\0\0" + out[8:]  # MTIME = time(NULL): masked
        wr[mode] = {"ret": zb.contents.ret, "bytes": out.hex(), "input": data.decode()}
        ref.zlibutil_buffer_free(zb)
    res["zlibutil_buffer_code_store"] = wr
    json.dump(res, open(os.path.join(HERE, "boundary.json"), "w"), indent=1)
    print("boundary.json:", json.dumps(res["hook_fastq_ff00"]))


def gen_full_flush(ref):
    """Run the reference's 7dictzip applet (oracle/_ref/cielbox_ref, built from the
    reference's own sources) on small inputs; a .dz is a gzip member whose RA extra
    field lists the chunk sizes (applet/7dictzip.c:300-330)."""
    import struct
    import subprocess
    import tempfile
    box = os.path.join(ROOT, "oracle", "_ref", "cielbox_ref")
    corpus = hdtest.corpus_small()
    out = []
    for cname, data in corpus.items():
        if not 0 < len(data) <= 6000:
            continue
        for name, level, opt in (("libdeflate", 1, "-cl1"), ("libdeflate", 6, "-cl6"), ("zlib", 6, "-cz6")):
            with tempfile.TemporaryDirectory() as td:
                fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.dz")
                open(fi, "wb").write(data)
                subprocess.run([box, "7dictzip", opt, fi, fo], check=True, capture_output=True)
                d = open(fo, "rb").read()
            assert d[:3] == b"\x1f\x8b\x08" and d[3] & 4
            xlen = struct.unpack("<H", d[10:12])[0]
            ex = d[12:12 + xlen]
            assert ex[:2] == b"RA"
            _, chlen, chcnt = struct.unpack("<HHH", ex[4:10])
            assert chcnt == 1
            clen = struct.unpack("<H", ex[10:12])[0]
            pos = 12 + xlen
            if d[3] & 8:
                pos = d.index(b"\0", pos) + 1
            chunk = d[pos:pos + clen]
            z = ref_encode(ref, name, level, data)
            out.append({"input": cname, "encoder": name, "level": level, "stream": b64(z), "flushed": b64(chunk)})
    json.dump(out, open(os.path.join(HERE, "full_flush.json"), "w"), indent=0)
    print("full_flush.json:", len(out), "chunks,", sum(len(o["flushed"]) for o in out) * 3 // 4, "bytes;",
          sum(1 for o in out if len(base64.b64decode(o["flushed"])) - len(base64.b64decode(o["stream"])) == 5),
          "with the extra zero byte")


RATIO_SETS = [("fastq", 201, 0xff00, 64), ("text", 202, 0xff00, 64), ("text", 203, 1 << 20, 4)]
RATIO_ENCODERS = [("libdeflate", 1), ("libdeflate", 2), ("libdeflate", 6), ("libdeflate", 9), ("zlib", 6), ("slz", 1)]


def ratio_input(kind, seed, block, nblocks):
    """the inputs of ratio_ref.json: `nblocks` blocks of `block` bytes cut from one seeded stream (regenerated by the
    tests from (kind, seed), not stored)"""
    s = hdtest.synth()
    n = block * nblocks
    return bytes(s.fastq_like(n, seed=seed) if kind == "fastq" else s.text_like(n, seed=seed))


def gen_ratio_ref(ref):
    """encoder parity = round trip + RATIO ENVELOPE (libdeflate's bytes are no golden,
    lib/libdeflate/libdeflate.h:75-83).  The compressed sizes the reference's encoders reach on seeded block sets:
    what tests/test_gpu_parity.py::test_ratio_envelope holds the kernels' sizes against."""
    out = []
    for kind, seed, block, nblocks in RATIO_SETS:
        data = ratio_input(kind, seed, block, nblocks)
        sizes = {}
        for name, level in RATIO_ENCODERS:
            tot = 0
            for b in range(nblocks):
                tot += len(ref_encode(ref, name, level, data[b * block:(b + 1) * block]))
            sizes["%s%d" % (name, level)] = tot
        out.append({"kind": kind, "seed": seed, "block": block, "nblocks": nblocks, "in_bytes": len(data),
                    "in_sha256": hdtest.sha(data), "ref_bytes": sizes})
        print("ratio_ref:", kind, block, {k: round(v / len(data), 4) for k, v in sizes.items()})
    json.dump(out, open(os.path.join(HERE, "ratio_ref.json"), "w"), indent=1)


def main():
    ref = hdtest.ref()
    assert ref is not None, "build the reference first: make -C oracle ref"
    if len(sys.argv) > 1 and sys.argv[1] == "ratio":
        return gen_ratio_ref(ref)
    gen_ratio_ref(ref)
    gen_std_vects(ref)
    gen_ref_streams(ref)
    gen_ref_streams_full(ref)
    gen_mutants(ref)
    gen_boundary(ref)
    gen_full_flush(ref)


if __name__ == "__main__":
    main()
