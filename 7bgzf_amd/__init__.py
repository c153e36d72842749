"""7bgzf_amd -- MI355X-native block-parallel DEFLATE behind 7bgzf's codec boundary.

This package is a thin ctypes view of ``libhipdeflate.so`` (C ABI declared in
``include/hipdeflate.h``; HIP kernels under ``7bgzf_amd/csrc``).  It exists for
the tests and ``bench.py``: the product is the shared library, which the
reference binds from C (see ``INTEGRATION.md``).  PyTorch is used only for
device memory, streams and ``torch.distributed``.

The directory name starts with a digit, so import it with
``importlib.import_module("7bgzf_amd")``.

There is no CPU fallback anywhere in here: a missing library raises on import of
:func:`lib`, and a missing/unsuitable GPU makes every call return
``HD_E_NODEVICE`` (raised as :class:`HipDeflateError`).
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libhipdeflate.so")

FRAME_RAW, FRAME_BGZF, FRAME_MIGZ, FRAME_RAW_FLUSH, FRAME_ZLIB, FRAME_GZIP = 0, 1, 2, 3, 4, 5
FRAME_LATENCY = 0x100        # OR'ed into a frame: several wavefronts per block (include/hipdeflate.h)
DEFLATE_HIP = 11
HD_E_NODEVICE, HD_E_ARG, HD_E_NOMEM = 100, 101, 102

BGZF_BLOCK = 0xff00          # applet/7bgzf.c:146-147, htslib BGZF_BLOCK_SIZE
WG_LEVEL = 3                 # include/hipdeflate_params.h HD_WG_LEVEL: levels >= this are the workgroup parse (one stream per block)
BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


class HipDeflateError(RuntimeError):
    pass


_lib = None
_vp = ctypes.c_void_p

# every symbol include/hipdeflate.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "hipdeflate_init", "hipdeflate_shutdown", "hipdeflate_available", "hipdeflate_version", "hipdeflate_stall_count",
    "hip_deflate", "hip_deflate_flush", "hip_inflate", "hipdeflate_batch_deflate", "hipdeflate_batch_inflate",
    "hipdeflate_batch_deflate_dev", "hipdeflate_batch_inflate_dev", "hipdeflate_scan_sizes_dev",
    "hipdeflate_compact_dev", "hipdeflate_scratch_bytes", "bgzf_compress", "hipdeflate_selftest",
    "hipdeflate_pipe_open", "hipdeflate_pipe_input", "hipdeflate_pipe_submit", "hipdeflate_pipe_result",
    "hipdeflate_pipe_close", "hipdeflate_unpipe_open", "hipdeflate_unpipe_input", "hipdeflate_unpipe_submit",
    "hipdeflate_unpipe_result", "hipdeflate_unpipe_close", "hipdeflate_test_build_lengths", "hipdeflate_test_beside",
    "hip_inflate_flush", "hipdeflate_batch_inflate_flush", "hipdeflate_batch_inflate_flush_dev", "hipdeflate_bound",
    "hipdeflate_compact_span_dev",
    "hipdeflate_init_devices", "hipdeflate_device_count", "hipdeflate_use_device",
    "hipdeflate_pipe_open_on", "hipdeflate_unpipe_open_on", "hipdeflate_lat_open_on",
    "hipdeflate_pipe_members", "hipdeflate_lat_open", "hipdeflate_lat_input", "hipdeflate_lat_run", "hipdeflate_lat_output", "hipdeflate_lat_close",
]


def _link_hip_runtime():
    """A process must hold exactly ONE HIP runtime.  PyTorch's wheel bundles its own
    libamdhip64.so; libhipdeflate.so's RPATH looks in 7bgzf_amd/hiprt/ first, so point
    that at torch's copy (same inode => the loader shares the instance whichever of
    the two is loaded first).  See 7bgzf_amd/csrc/Makefile, target `hiprt`."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    link = os.path.join(HERE, "hiprt", "libamdhip64.so.7")
    target = None
    if spec is not None and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            target = cand
    try:
        if target is None:
            if os.path.islink(link) or os.path.exists(link):
                os.remove(link)
        elif not (os.path.islink(link) and os.readlink(link) == target):
            os.makedirs(os.path.dirname(link), exist_ok=True)
            if os.path.islink(link) or os.path.exists(link):
                os.remove(link)
            os.symlink(target, link)
    except OSError:
        pass


def lib():
    """Load libhipdeflate.so (built by ``__graft_entry__.build()``); loud if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipDeflateError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C 7bgzf_amd/csrc). There is no CPU fallback." % LIB_PATH)
    _link_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    L.hipdeflate_version.restype = ctypes.c_char_p
    L.hipdeflate_scratch_bytes.restype = ctypes.c_uint64
    L.hipdeflate_bound.restype = ctypes.c_uint64
    L.hipdeflate_bound.argtypes = [ctypes.c_uint64, ctypes.c_int]
    L.hipdeflate_init.argtypes = [ctypes.c_int]
    L.hipdeflate_stall_count.restype = ctypes.c_uint64
    L.hipdeflate_stall_count.argtypes = []
    L.hipdeflate_init_devices.argtypes = [_vp, ctypes.c_int]
    L.hipdeflate_use_device.argtypes = [ctypes.c_int]
    L.hipdeflate_pipe_open_on.restype = _vp
    L.hipdeflate_pipe_open_on.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int]
    L.hipdeflate_unpipe_open_on.restype = _vp
    L.hipdeflate_unpipe_open_on.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
    L.hipdeflate_lat_open_on.restype = _vp
    L.hipdeflate_lat_open_on.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
    sz_p = ctypes.POINTER(ctypes.c_size_t)
    L.hip_deflate.argtypes = [_vp, sz_p, _vp, ctypes.c_size_t, ctypes.c_int]
    L.hip_deflate_flush.argtypes = [_vp, sz_p, _vp, ctypes.c_size_t, ctypes.c_int]
    L.hip_inflate.argtypes = [_vp, sz_p, _vp, ctypes.c_size_t]
    L.hip_inflate_flush.argtypes = [_vp, sz_p, _vp, ctypes.c_size_t]
    L.bgzf_compress.argtypes = [_vp, sz_p, _vp, ctypes.c_size_t, ctypes.c_int]
    L.hipdeflate_batch_deflate.argtypes = [_vp, _vp, _vp, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, _vp,
                                           ctypes.c_uint64, ctypes.c_uint32, _vp, _vp, _vp]
    L.hipdeflate_batch_inflate.argtypes = [_vp, _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]
    L.hipdeflate_batch_inflate_flush.argtypes = L.hipdeflate_batch_inflate.argtypes
    L.hipdeflate_batch_deflate_dev.argtypes = [_vp, _vp, _vp, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, _vp,
                                               ctypes.c_uint64, ctypes.c_uint32, _vp, _vp, _vp, _vp]
    L.hipdeflate_batch_inflate_dev.argtypes = [_vp, _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.hipdeflate_batch_inflate_flush_dev.argtypes = L.hipdeflate_batch_inflate_dev.argtypes
    L.hipdeflate_scan_sizes_dev.argtypes = [_vp, ctypes.c_uint32, ctypes.c_uint64, _vp, _vp, _vp]
    L.hipdeflate_compact_dev.argtypes = [_vp, ctypes.c_uint64, _vp, _vp, ctypes.c_uint32, _vp, _vp]
    L.hipdeflate_compact_span_dev.argtypes = [_vp, ctypes.c_uint64, _vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint64, _vp]
    L.hipdeflate_pipe_open.restype = _vp
    L.hipdeflate_pipe_open.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int]
    L.hipdeflate_pipe_input.restype = _vp
    L.hipdeflate_pipe_input.argtypes = [_vp, sz_p]
    L.hipdeflate_pipe_submit.argtypes = [_vp, ctypes.c_size_t]
    L.hipdeflate_pipe_result.argtypes = [_vp, ctypes.POINTER(_vp), sz_p, ctypes.POINTER(ctypes.c_uint32)]
    L.hipdeflate_pipe_members.argtypes = [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp)]
    L.hipdeflate_pipe_close.restype = None
    L.hipdeflate_pipe_close.argtypes = [_vp]
    L.hipdeflate_unpipe_open.restype = _vp
    L.hipdeflate_unpipe_open.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
    L.hipdeflate_unpipe_input.restype = _vp
    L.hipdeflate_unpipe_input.argtypes = [_vp, sz_p]
    L.hipdeflate_unpipe_submit.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_uint32]
    L.hipdeflate_unpipe_result.argtypes = [_vp, ctypes.POINTER(_vp), sz_p]
    L.hipdeflate_unpipe_close.restype = None
    L.hipdeflate_unpipe_close.argtypes = [_vp]
    L.hipdeflate_lat_open.restype = _vp
    L.hipdeflate_lat_open.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
    L.hipdeflate_lat_input.restype = _vp
    L.hipdeflate_lat_input.argtypes = [_vp, ctypes.c_uint32]
    L.hipdeflate_lat_run.argtypes = [_vp, _vp, ctypes.c_uint32]
    L.hipdeflate_lat_output.restype = _vp
    L.hipdeflate_lat_output.argtypes = [_vp, ctypes.c_uint32, _vp, _vp, _vp]
    L.hipdeflate_lat_close.restype = None
    L.hipdeflate_lat_close.argtypes = [_vp]
    L.hipdeflate_test_build_lengths.argtypes = [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, _vp]
    L.hipdeflate_test_beside.argtypes = [ctypes.c_int, ctypes.c_uint32]
    L.hipdeflate_test_beside.restype = None
    _lib = L
    return L


def _check(rc, what):
    if rc:
        raise HipDeflateError("%s failed with %d%s" % (
            what, rc, " (no usable gfx950 device; there is no CPU fallback)" if rc == HD_E_NODEVICE else ""))


def available():
    """True iff a usable MI355X is present and the kernels loaded."""
    return lib().hipdeflate_available() == 0


def version():
    lib().hipdeflate_available()
    return lib().hipdeflate_version().decode()


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(_vp)


def as_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(0, dtype=np.uint8)


# ---- per-block codecs (zlibutil_code_enc / _dec, lib/zlibutil.h:46-47) -----------


def hip_deflate(data, level=1, cap=None):
    """-> (ret, bytes).  Mirrors libdeflate_deflate's contract (lib/zlibutil.c:179)."""
    src = as_u8(data)
    cap = (len(src) + len(src) // 2 + 64) if cap is None else cap
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    n = ctypes.c_size_t(cap)
    r = lib().hip_deflate(_p(dst), ctypes.byref(n), _p(src), len(src), level)
    return r, bytes(dst[: n.value]) if r == 0 else b""


def hip_deflate_flush(data, level=1, cap=None):
    """-> (ret, bytes) in full-flush form: what zlibutil_buffer_full_flush
    (applet/7dictzip.c:93-126) makes of a codec's output."""
    src = as_u8(data)
    cap = (len(src) + len(src) // 2 + 64) if cap is None else cap
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    n = ctypes.c_size_t(cap)
    r = lib().hip_deflate_flush(_p(dst), ctypes.byref(n), _p(src), len(src), level)
    return r, bytes(dst[: n.value]) if r == 0 else b""


def hip_inflate(data, cap):
    """-> (ret, bytes).  Mirrors libdeflate_inflate's contract (lib/zlibutil.c:194)."""
    src = as_u8(data)
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    n = ctypes.c_size_t(cap)
    r = lib().hip_inflate(_p(dst), ctypes.byref(n), _p(src), len(src))
    return r, bytes(dst[: n.value]) if r == 0 else b""


def hip_inflate_flush(data, cap):
    """-> (ret, bytes) for a full-flushed chunk (no final block): the decoder side of
    hip_deflate_flush, as zlib_inflate / igzip_inflate in applet/7dictzip.c:318-323."""
    src = as_u8(data)
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    n = ctypes.c_size_t(cap)
    r = lib().hip_inflate_flush(_p(dst), ctypes.byref(n), _p(src), len(src))
    return r, bytes(dst[: n.value]) if r == 0 else b""


def pipe_compress(data, level=1, frame=FRAME_BGZF, block=0xff00, per_batch=64, depth=3):
    """Run `data` through the streaming encoder (hipdeflate_pipe_*): as many batches in flight as the
    pipe allows before results are fetched.  -> the concatenated members (no EOF member)."""
    L = lib()
    src = as_u8(data)
    p = L.hipdeflate_pipe_open(level, frame, block, per_batch, depth)
    if not p:
        raise HipDeflateError("hipdeflate_pipe_open failed")
    out, pos, inflight = [], 0, 0

    def fetch():
        d, n, nb = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_uint32()
        r = L.hipdeflate_pipe_result(p, ctypes.byref(d), ctypes.byref(n), ctypes.byref(nb))
        if r:
            raise HipDeflateError("hipdeflate_pipe_result %d" % r)
        out.append(ctypes.string_at(d, n.value))

    try:
        while True:
            if inflight == depth - 1:        # one slot stays with the caller (the held result)
                fetch()
                inflight -= 1
            cap = ctypes.c_size_t()
            buf = L.hipdeflate_pipe_input(p, ctypes.byref(cap))
            if not buf:
                raise HipDeflateError("hipdeflate_pipe_input failed")
            n = min(cap.value, len(src) - pos)
            if n:
                ctypes.memmove(buf, src[pos:pos + n].ctypes.data, n)
            if L.hipdeflate_pipe_submit(p, n):
                raise HipDeflateError("hipdeflate_pipe_submit failed")
            inflight += 1
            pos += n
            if n < cap.value:
                break
        while inflight:
            fetch()
            inflight -= 1
    finally:
        L.hipdeflate_pipe_close(p)
    return b"".join(out)


def bgzf_compress_hook(data, cap=0x10000):
    """Call the exported LD_PRELOAD hook (bgzf_compress.c:39). -> (ret, member bytes)."""
    src = as_u8(data)
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    n = ctypes.c_size_t(cap)
    r = lib().bgzf_compress(_p(dst), ctypes.byref(n), _p(src) if len(src) else None, len(src), -1)
    return r, bytes(dst[: n.value]) if r == 0 else b""


# ---- batch API, host buffers ---------------------------------------------------------


def split_blocks(n, block_size):
    offs = np.arange(0, max(n, 1), block_size, dtype=np.uint64) if n else np.zeros(0, dtype=np.uint64)
    lens = np.minimum(n - offs.astype(np.int64), block_size).astype(np.uint32) if n else np.zeros(0, dtype=np.uint32)
    return offs, lens


def batch_deflate(data, offs, lens, level=1, frame=FRAME_RAW, slot=None):
    """Compress blocks data[offs[i]:offs[i]+lens[i]].  -> (members list, crc32 array, status array)."""
    src = as_u8(data)
    offs = _np(offs, np.uint64)
    lens = _np(lens, np.uint32)
    nb = len(offs)
    if slot is None:
        mx = int(lens.max()) if nb else 0
        slot = int(lib().hipdeflate_bound(mx, level))
    out = np.zeros(max(nb * slot, 1), dtype=np.uint8)
    olen = np.zeros(nb, dtype=np.uint32)
    crc = np.zeros(nb, dtype=np.uint32)
    st = np.zeros(nb, dtype=np.int32)
    _check(lib().hipdeflate_batch_deflate(_p(src), _p(offs), _p(lens), nb, level, frame, _p(out), slot, slot,
                                          _p(olen), _p(crc), _p(st)), "hipdeflate_batch_deflate")
    members = [bytes(out[i * slot: i * slot + int(olen[i])]) for i in range(nb)]
    return members, crc, st


def batch_inflate(streams, caps, want_crc=True, flushed=False):
    """Inflate a list of raw-DEFLATE streams.  -> (outputs list, crc32 array, status array).
    flushed: the streams are full-flushed chunks (hipdeflate_batch_inflate_flush)."""
    nb = len(streams)
    ilen = np.array([len(s) for s in streams], dtype=np.uint32)
    ioff = np.zeros(nb, dtype=np.uint64)
    if nb:
        ioff[1:] = np.cumsum(ilen[:-1], dtype=np.uint64)
    src = as_u8(b"".join(bytes(s) for s in streams))
    caps = _np(caps, np.uint32)
    ooff = np.zeros(nb, dtype=np.uint64)
    if nb:
        ooff[1:] = np.cumsum(caps[:-1].astype(np.uint64))
    out = np.zeros(max(int(caps.astype(np.uint64).sum()), 1), dtype=np.uint8)
    olen = np.zeros(nb, dtype=np.uint32)
    crc = np.zeros(nb, dtype=np.uint32)
    st = np.zeros(nb, dtype=np.int32)
    fn = lib().hipdeflate_batch_inflate_flush if flushed else lib().hipdeflate_batch_inflate
    _check(fn(_p(src), _p(ioff), _p(ilen), nb, _p(out), _p(ooff), _p(caps), _p(olen),
              _p(crc) if want_crc else None, _p(st)), "hipdeflate_batch_inflate")
    outs = [bytes(out[int(ooff[i]): int(ooff[i]) + int(olen[i])]) if st[i] == 0 else b"" for i in range(nb)]
    return outs, crc, st


# ---- container level (role of applet/7bgzf.c _compress / _decompress) --------------


def bgzf_compress_bytes(data, level=1, block_size=BGZF_BLOCK):
    """Whole-buffer BGZF writer: members in order + the 28-byte EOF member
    (applet/7bgzf.c:159-289 with the per-block loop turned into one batch)."""
    src = as_u8(data)
    offs, lens = split_blocks(len(src), block_size)
    members, _, st = batch_deflate(src, offs, lens, level, FRAME_BGZF, slot=65536)
    if np.any(st != 0):
        raise HipDeflateError("hip_deflate %d" % int(st[np.nonzero(st)[0][0]]))
    return b"".join(members) + BGZF_EOF


def bgzf_scan(blob):
    """Pre-scan the BSIZE chain (the serial part of applet/7bgzf.c:306-328).
    -> list of (payload_offset, payload_len_incl_trailer, isize)."""
    out = []
    p = 0
    n = len(blob)
    while p < n:
        if n - p < 18 or blob[p] != 0x1f or blob[p + 1] != 0x8b or blob[p + 3] != 4 or blob[p + 12:p + 16] != b"BC\x02\x00":
            raise HipDeflateError("not BGZF or corrupted")
        total = int.from_bytes(blob[p + 16:p + 18], "little") + 1
        if p + total > n:
            raise HipDeflateError("not BGZF or corrupted")
        isize = int.from_bytes(blob[p + total - 4:p + total], "little")
        out.append((p + 18, total - 18, isize))
        p += total
    return out


def unpipe_decompress(blob, members_per_batch=16, depth=3):
    """BGZF bytes through the streaming decoder (hipdeflate_unpipe_*), `members_per_batch` members per
    batch, as many batches in flight as the pipe allows.  -> (status of the first failing batch or 0, output)."""
    L = lib()
    blob = bytes(blob)
    tbl = bgzf_scan(blob)
    groups = [tbl[i:i + members_per_batch] for i in range(0, len(tbl), members_per_batch)] or [[]]
    in_cap = max([g[-1][0] + g[-1][1] - (g[0][0] - 18) for g in groups if g] + [64])
    out_cap = max([sum(m[2] for m in g) for g in groups] + [64])
    p = L.hipdeflate_unpipe_open(members_per_batch, in_cap, out_cap, depth)
    if not p:
        raise HipDeflateError("hipdeflate_unpipe_open failed")
    out, inflight, status = [], 0, 0

    def fetch():
        nonlocal status
        d, n = ctypes.c_void_p(), ctypes.c_size_t()
        r = L.hipdeflate_unpipe_result(p, ctypes.byref(d), ctypes.byref(n))
        if r and not status:
            status = r
        out.append(ctypes.string_at(d, n.value) if not r else b"")

    try:
        for g in groups:
            if inflight == depth - 1:
                fetch()
                inflight -= 1
            cap = ctypes.c_size_t()
            buf = L.hipdeflate_unpipe_input(p, ctypes.byref(cap))
            if not buf:
                raise HipDeflateError("hipdeflate_unpipe_input failed")
            base = g[0][0] - 18 if g else 0
            end = g[-1][0] + g[-1][1] if g else 0
            if end > base:
                ctypes.memmove(buf, blob[base:end], end - base)
            ioff = np.array([m[0] - base for m in g], dtype=np.uint64)
            ilen = np.array([m[1] for m in g], dtype=np.uint32)
            osz = np.array([m[2] for m in g], dtype=np.uint32)
            if L.hipdeflate_unpipe_submit(p, _p(ioff) if len(g) else None, _p(ilen) if len(g) else None,
                                          _p(osz) if len(g) else None, len(g)):
                raise HipDeflateError("hipdeflate_unpipe_submit failed")
            inflight += 1
        while inflight:
            fetch()
            inflight -= 1
    finally:
        L.hipdeflate_unpipe_close(p)
    return status, b"".join(out)


def bgzf_decompress_bytes(blob, verify=True):
    """Whole-buffer BGZF reader (applet/7bgzf.c:295-365): every member's payload +
    8-byte trailer is handed to the inflater (``:328``), capacity = ISIZE."""
    blob = bytes(blob)
    tbl = bgzf_scan(blob)
    streams = [blob[o:o + ln] for o, ln, _ in tbl]
    caps = [isz for _, _, isz in tbl]
    outs, crc, st = batch_inflate(streams, caps)
    for i, (o, ln, isz) in enumerate(tbl):
        if st[i] != 0:
            raise HipDeflateError("inflate %d" % int(st[i]))
        if verify:
            want = int.from_bytes(blob[o + ln - 8:o + ln - 4], "little")
            if len(outs[i]) != isz or int(crc[i]) != want:
                raise HipDeflateError("member %d: CRC32/ISIZE mismatch" % i)
    return b"".join(outs)
