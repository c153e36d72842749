"""Device-resident view of the batch API for bench.py and the GPU tests: torch
tensors own the HBM buffers (torch is plumbing here -- allocation, streams,
torch.distributed), libhipdeflate.so does the work on the current stream."""
import importlib

import numpy as np
import torch

_pkg = importlib.import_module(__package__)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class DeviceDeflate:
    """Pre-allocated buffers for compressing `nblocks` blocks that live in HBM."""

    def __init__(self, nblocks, slot=65536, device="cuda"):
        self.nblocks = nblocks
        self.slot = slot
        self.slots = torch.empty(nblocks * slot, dtype=torch.uint8, device=device)
        self.out_len = torch.zeros(nblocks, dtype=torch.int32, device=device)
        self.crc = torch.zeros(nblocks, dtype=torch.int32, device=device)
        self.status = torch.zeros(nblocks, dtype=torch.int32, device=device)
        self.dst_off = torch.zeros(nblocks, dtype=torch.int64, device=device)
        self.total = torch.zeros(1, dtype=torch.int64, device=device)

    def run(self, data, in_off, in_len, level=1, frame=_pkg.FRAME_BGZF):
        rc = _pkg.lib().hipdeflate_batch_deflate_dev(
            _ptr(data), _ptr(in_off), _ptr(in_len), self.nblocks, level, frame, _ptr(self.slots), self.slot,
            self.slot, _ptr(self.out_len), _ptr(self.crc), _ptr(self.status), _stream())
        _pkg._check(rc, "hipdeflate_batch_deflate_dev")

    def scan(self, base=0):
        rc = _pkg.lib().hipdeflate_scan_sizes_dev(_ptr(self.out_len), self.nblocks, base, _ptr(self.dst_off),
                                                  _ptr(self.total), _stream())
        _pkg._check(rc, "hipdeflate_scan_sizes_dev")

    def compact(self, dst, span_base=0):
        """members -> dst; with span_base != 0 dst is this rank's span of a sharded stream and
        dst_off[] (scan(base=span_base)) are offsets in the whole stream"""
        rc = _pkg.lib().hipdeflate_compact_span_dev(_ptr(self.slots), self.slot, _ptr(self.out_len),
                                                    _ptr(self.dst_off), self.nblocks, _ptr(dst), span_base, _stream())
        _pkg._check(rc, "hipdeflate_compact_span_dev")


def device_inflate(comp, in_off, in_len, out, out_off, out_cap, out_len, crc, status):
    nb = in_off.numel()
    rc = _pkg.lib().hipdeflate_batch_inflate_dev(_ptr(comp), _ptr(in_off), _ptr(in_len), nb, _ptr(out), _ptr(out_off),
                                                 _ptr(out_cap), _ptr(out_len), _ptr(crc), _ptr(status), _stream())
    _pkg._check(rc, "hipdeflate_batch_inflate_dev")


def block_table(total_bytes, block_size, device="cuda"):
    nb = (total_bytes + block_size - 1) // block_size
    off = torch.arange(nb, dtype=torch.int64, device=device) * block_size
    ln = torch.clamp(total_bytes - off, max=block_size).to(torch.int32)
    return off, ln


def to_numpy_u32(t):
    return t.cpu().numpy().view(np.uint32)
