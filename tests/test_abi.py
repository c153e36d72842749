"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/hipdeflate.h declares, and refuses to work without a GPU instead of
silently falling back to a CPU codec."""
import ctypes
import os
import re
import subprocess

import pytest

import hdtest


@pytest.fixture(scope="module")
def pkg():
    p = hdtest.pkg()
    if not os.path.exists(p.LIB_PATH):
        subprocess.run(["make", "-s", "-C", os.path.join(hdtest.ROOT, "7bgzf_amd", "csrc")], check=True)
    return p


def declared_functions():
    text = open(os.path.join(hdtest.ROOT, "include", "hipdeflate.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hip_\w+|hipdeflate_\w+|bgzf_compress)\s*\(", text)))


def test_exports_match_header(pkg):
    names = declared_functions()
    assert len(names) >= 15
    assert sorted(pkg.EXPORTS) == names
    lib = pkg.lib()
    for n in names:
        assert getattr(lib, n) is not None
    out = subprocess.run(["nm", "-D", "--defined-only", pkg.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    for n in names:
        assert n in exported, n


def test_no_oracle_or_reference_in_product(pkg):
    """The product must not link or embed the oracle / the reference."""
    out = subprocess.run(["readelf", "-d", pkg.LIB_PATH], capture_output=True, text=True).stdout
    needed = re.findall(r"NEEDED.*\[(.*)\]", out)          # direct dependencies only
    assert needed and all(not n.startswith(("liboracle", "libref", "libdeflate", "libz.")) for n in needed), needed
    src_dir = os.path.join(hdtest.ROOT, "7bgzf_amd")
    for root, _, files in os.walk(src_dir):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".hpp")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "liboracle" not in text and "hd_oracle.h" not in text and "libref.so" not in text, f


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present: covered by the gpu tests")
def test_fails_loudly_without_gpu(pkg):
    assert pkg.available() is False
    r, z = pkg.hip_deflate(b"hello hello hello hello")
    assert r == pkg.HD_E_NODEVICE and z == b""
    r, z = pkg.hip_inflate(b"\x03\x00", 10)
    assert r == pkg.HD_E_NODEVICE
    with pytest.raises(pkg.HipDeflateError):
        pkg.bgzf_compress_bytes(b"abc")


def test_device_list_from_the_environment_survives_init_minus_one(pkg):
    """hipdeflate_init(-1) -- what every host tool calls first -- reads HIPDEFLATE_DEVICES like the lazy entry points do
    (it used to pin a list of one, so `hd7bgzf -g 2` under HIPDEFLATE_DEVICES=0,0 put both pipes on entry 0);
    an explicit ordinal is still a list of one.  No GPU needed: the list is configured before a context is made."""
    child = ("import ctypes,sys; L=ctypes.CDLL(%r); r=L.hipdeflate_init(int(sys.argv[1])); "
             "print('COUNT', L.hipdeflate_device_count())") % pkg.LIB_PATH
    for arg, env, want in (("-1", {"HIPDEFLATE_DEVICES": "0,0"}, 2), ("-1", {"HIPDEFLATE_DEVICES": "0,1,2"}, 3),
                           ("-1", {}, 1), ("0", {"HIPDEFLATE_DEVICES": "0,0"}, 1)):
        e = {k: v for k, v in os.environ.items() if k != "HIPDEFLATE_DEVICES"}
        e.update(env)
        p = subprocess.run([__import__("sys").executable, "-c", child, arg], env=e, capture_output=True, text=True, timeout=120)
        assert "COUNT %d" % want in p.stdout, (arg, env, p.stdout, p.stderr[-500:])


def test_hook_constants_without_gpu(pkg):
    """slen == 0 -> canned EOF member; too-small capacity -> -1 (bgzf_compress.c:40-51)."""
    import json
    b = json.load(open(os.path.join(hdtest.GOLDEN, "boundary.json")))
    r, m = pkg.bgzf_compress_hook(b"", 0x10000)
    assert r == b["hook_eof"]["ret"] == 0 and m.hex() == b["hook_eof"]["member"] and m == pkg.BGZF_EOF
    r, _ = pkg.bgzf_compress_hook(b"", 27)
    assert r == b["hook_eof_cap27"]["ret"] == -1


def test_zlibutil_mirror_wrappers(pkg):
    """hd_zlibutil_buffer_code with a CPU-side stand-in codec (the store_deflate layout
    is irrelevant here: the wrappers only add bytes around whatever func returns) must
    produce the reference's RFC1950 / RFC1952 bytes (lib/zlibutil.c:374-405)."""
    import json
    b = json.load(open(os.path.join(hdtest.GOLDEN, "boundary.json")))["zlibutil_buffer_code_store"]
    lib = pkg.lib()

    class ZB(ctypes.Structure):
        _fields_ = [("dest", ctypes.c_void_p), ("destLen", ctypes.c_size_t), ("source", ctypes.c_void_p),
                    ("sourceLen", ctypes.c_size_t), ("func", ctypes.c_void_p), ("encode", ctypes.c_int),
                    ("level", ctypes.c_int), ("rfc1950", ctypes.c_int), ("rfc1952", ctypes.c_int),
                    ("ret", ctypes.c_int)]

    lib.hd_zlibutil_buffer_allocate.restype = ctypes.POINTER(ZB)
    lib.hd_zlibutil_buffer_allocate.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
    lib.hd_zlibutil_buffer_code.restype = ctypes.POINTER(ZB)
    lib.hd_zlibutil_buffer_code.argtypes = [ctypes.POINTER(ZB)]
    lib.hd_zlibutil_buffer_free.argtypes = [ctypes.POINTER(ZB)]
    ENC = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p,
                           ctypes.c_size_t, ctypes.c_int)

    def stored_codec(dest, dest_len, source, source_len, level):
        data = ctypes.string_at(source, source_len)
        n = len(data)
        z = bytes([1, n & 0xff, n >> 8, ~n & 0xff, (~n >> 8) & 0xff]) + data
        ctypes.memmove(dest, z, len(z))
        dest_len[0] = len(z)
        return 0

    cb = ENC(stored_codec)
    for mode in ("rfc1950", "rfc1952"):
        data = b[mode]["input"].encode()
        zb = lib.hd_zlibutil_buffer_allocate(200, len(data))
        ctypes.memmove(zb.contents.source, data, len(data))
        zb.contents.func = ctypes.cast(cb, ctypes.c_void_p)
        zb.contents.encode = 1
        setattr(zb.contents, mode, 1)
        ret = lib.hd_zlibutil_buffer_code(zb)
        assert ctypes.addressof(ret.contents) == ctypes.addressof(zb.contents)  # returns its argument
        out = ctypes.string_at(zb.contents.dest, zb.contents.destLen)
        if mode == "rfc1952":
            out = out[:4] + b"\0\0\0\0" + out[8:]
        assert zb.contents.ret == b[mode]["ret"] == 0 and out.hex() == b[mode]["bytes"], mode
        lib.hd_zlibutil_buffer_free(zb)
    # host checksums used by those wrappers
    lib.hd_crc32.restype = ctypes.c_uint32
    lib.hd_adler32.restype = ctypes.c_uint32
    v = b"123456789" * 1000
    assert lib.hd_crc32(0, v, len(v)) == hdtest.oracle_crc32(v)
    a = hdtest.as_u8(v)
    assert lib.hd_adler32(1, v, len(v)) == hdtest.oracle().hdo_adler32(1, a.ctypes.data, len(a))


def test_kernel_resource_budgets():
    """The compiler's resource report of the build (7bgzf_amd/csrc/hd_api.resources.log, written by the
    Makefile): no kernel may spill, and the dynamic-level kernels must stay at <= 168 VGPRs -- their
    persistent grids (dynamic_grid(), hd_deflate_dynamic.hpp) assume three waves per SIMD; one
    register more and a third of the grid runs as a serial tail (measured: 96 -> 55 GB/s)."""
    import re
    log = os.path.join(os.path.dirname(hdtest.pkg().LIB_PATH), "csrc", "hd_api.resources.log")
    assert os.path.exists(log), "build with make -C 7bgzf_amd/csrc"
    kernels, cur = {}, None
    for line in open(log):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    # k_deflate_dynamic<W, H, MINLEN, LAZY, EMIT, INTRA, DEEP, PARTS>: the emit-only instantiations have EMIT = 1 (PARTS = 0, and
    # HD_LAT_PARTS_MAX for the latency segments parsed in parts)
    emit = {k: v for k, v in kernels.items() if re.search(r"k_deflate_dynamicILi\d+ELi\d+ELi\d+ELi\d+ELi1ELi\d+ELi\d+ELi\d+ELi\d+EEEv", k)}
    dyn = {k: v for k, v in kernels.items() if "k_deflate_dynamic" in k and k not in emit}
    sta = {k: v for k, v in kernels.items() if "k_deflate_static" in k}
    inf = {k: v for k, v in kernels.items() if "k_inflateE" in k}
    inf_lat = {k: v for k, v in kernels.items() if "k_inflate_lat" in k}
    wg = {k: v for k, v in kernels.items() if "k_parse_wg" in k}
    # (round 4: levels 6..9 are the workgroup parse in the throughput form; their latency segments share level 6's two-way
    # kernels, so the 16 / 32 KiB geometries of rounds 2-3 are gone)
    # (round 4, later: levels 3..9 reach the one-wavefront kernels only for their latency segments, parsed in parts: their fused
    # kernels are not instantiated any more -- the fused kernel is level 2's)
    # (round 5: levels 3..9 are the workgroup parse in EVERY form -- one codec per level --, their member written by the one-wavefront
    # emit kernel or, behind the per-block boundary, by a workgroup (k_emit_wg); the one-wavefront parse kernels left are level 1's
    # (plain and primed) and level 2's)
    emit_wg = {k: v for k, v in kernels.items() if "k_emit_wg" in k}
    assert len(dyn) == 1 and len(emit) == 3 and len(sta) == 3 and len(inf) == 1 and len(inf_lat) == 1 and len(wg) == 8 and len(emit_wg) == 1, list(kernels)
    (v,) = emit_wg.values()
    assert v["VGPRs"] <= 128 and v["LDS Size"] <= 163840, v      # sixteen wavefronts, one workgroup per CU
    for k, v in kernels.items():
        # (the one exception: the emit-only kernel's BESIDE instantiation spills a few of its ~230 scalar values past the VGPR lanes --
        # 32 bytes per lane, touched at block boundaries; it must stay within 128 registers to fit beside the parse, see below)
        assert v["ScratchSize"] == 0 or (k.endswith("ELi0ELi1EEEvNS_11DeflateArgsE") and "k_deflate_dynamic" in k and v["ScratchSize"] <= 64), (k, v)
    for k, v in dyn.items():
        assert v["VGPRs"] <= 168, (k, v)
        # LDS is granted in 1280-byte units (measured: 10 waves of 15584 B do not fit a CU, of 15328 B do)
        units = -(-v["LDS Size"] // 1280)
        deep = k.endswith("ELi1ELi0EEEvNS_11DeflateArgsE")      # the two-way tables (levels 6..9)
        want = 12 if ("Li13ELi11E" in k or "Li12ELi11E" in k) else (18 if deep else 14) if "Li13ELi12E" in k else \
            25 if "Li14ELi12E" in k else 32 if "Li14ELi13E" in k else 48
        assert units <= want, (k, v)                 # 10 / 9 / 7 / 5 / 4 / 2 waves per CU: dynamic_grid()
    for k, v in emit.items():
        # 16 waves per CU: launch_level().  The PARTS instantiation (latency segments: a handful of workgroups on an empty
        # chip) is two wavefronts and two construction scratches per workgroup
        parts = re.search(r"ELi0ELi[01]EEEvNS_11DeflateArgsE$", k) is None      # (..., PARTS, BESIDE)
        assert v["VGPRs"] <= 128 and v["LDS Size"] <= (13 if parts else 8) * 1280, (k, v)
    for k, v in sta.items():                                     # level 1 and the parse kernels of levels 2-9
        # the level-1 geometry runs 18 waves per CU = five per SIMD on two of them: <= 96 VGPRs; the two-way parse kernels
        # (8 / 5 / 4 waves per CU) have 168; the others <= 128 (four per SIMD)
        # k_deflate_static<W, H, TOK, MINLEN, LAZY, INTRA, DEEP, PRIMED>
        deep = re.search(r"ELi1ELb[01]EEEvNS_11DeflateArgsE$", k) is not None
        assert v["VGPRs"] <= (96 if "Li12ELi11E" in k else 168 if deep else 128), (k, v)
        units = -(-v["LDS Size"] // 1280)
        tok = "ELb1E" in k                           # parse kernels; level 1 itself is ELb0E
        want = 7 if "Li12ELi11E" in k else 10 if "Li13ELi11E" in k else (16 if deep else 12) if "Li13ELi12E" in k else \
            25 if "Li14ELi12E" in k else 32 if "Li14ELi13E" in k else 48
        assert units <= want, (k, v)                 # 18 / 12 / 10 / 8 / 5 / 4 / 2 waves per CU: parse_slots() (two-way tables from level 6 on)
    (v,) = inf.values()
    assert v["VGPRs"] <= 80 and v["LDS Size"] <= 6400, v         # five LDS units (25 per CU), 6 waves per SIMD: 24 waves per CU
    (v,) = inf_lat.values()
    # the whole window in LDS + the queues and the spec ring of its four wavefronts: ONE workgroup per CU (a stream wants the CU's
    # four SIMDs for its four wavefronts; a batch holds at most 64 streams and two batches are out at once: 128 workgroups, 256 CUs)
    assert v["VGPRs"] <= 128 and 81920 < v["LDS Size"] <= 98304, v
    # k_parse_wg<WAYS, LAZY> (levels 3 / 4 / 5 / 6..9): one workgroup of 16 wavefronts per CU: four per SIMD (<= 128 VGPRs),
    # ring + table + state within a CU's 160 KiB
    for k, v in wg.items():
        if k.endswith("ELi1EEEvNS_11DeflateArgsE"):
            # BESIDE (round 5): the parse's 131 KB of LDS are passed at launch -- "LDS Size" 0 here -- and it holds at most 96 registers,
            # so that four of its wavefronts and one emit wavefront of up to 128 fit a SIMD's 512 (hd_deflate_wg.hpp)
            assert v["VGPRs"] <= 96 and v["LDS Size"] == 0, (k, v)
        else:
            assert v["VGPRs"] <= 128 and v["LDS Size"] <= 163840, (k, v)
    for k, v in emit.items():
        if k.endswith("ELi0ELi1EEEvNS_11DeflateArgsE"):
            assert v["VGPRs"] <= 128 and v["LDS Size"] <= 8 * 1280, (k, v)      # ... the emit-only kernel that runs beside it


def test_container_hosts_crc_fold_matches_zlib(tmp_path):
    """hd7dictzip / hd7razf write a member's CRC-32 folded from the kernel's per-chunk CRCs
    (7bgzf_amd/csrc/hd_host_util.h: crc(A||B) = crc(A) * x^(8|B|) + crc(B) in GF(2)[x]/P).  The fold itself needs
    no GPU: compiled here and compared with zlib.crc32 over random splits, equal and ragged chunk sizes."""
    import subprocess
    import zlib
    import numpy as np
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "crc_fold_check.c")
    inc = os.path.join(hdtest.ROOT, "7bgzf_amd", "csrc")
    exe = str(tmp_path / "crc_fold_check")
    subprocess.run(["gcc", "-O2", "-I", inc, "-o", exe, src], check=True)
    rng = np.random.default_rng(3)
    for trial in range(12):
        data = rng.integers(0, 256, int(rng.integers(1, 400000)), dtype=np.uint8).tobytes()
        if trial % 3 == 0:
            cuts = list(range(0, len(data), 58315)) + [len(data)]                 # dictzip's chunking
        elif trial % 3 == 1:
            cuts = list(range(0, len(data), 32768)) + [len(data)]                 # razf's
        else:
            cuts = sorted(set([0, len(data)] + [int(x) for x in rng.integers(0, len(data) + 1, 9)]))
        parts = [data[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
        text = "".join("%08x %d\n" % (zlib.crc32(p), len(p)) for p in parts)
        out = subprocess.run([exe], input=text.encode(), capture_output=True, check=True).stdout.decode().strip()
        assert int(out, 16) == zlib.crc32(data), (trial, len(parts))
