"""The N > 1 path on CPU: two gloo ranks shard a stream by block range, exchange
their compressed totals with ONE all_gather and place their spans at the derived
base offsets; the assembled file must equal the single-rank stream byte for byte.
(The per-block codec here is the oracle's CPU twin -- the point of this test is
the sharding / offset logic, which is the same code bench.py runs over RCCL.)"""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

import hdtest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, hdtest.ROOT)
    sys.path.insert(0, os.path.join(hdtest.ROOT, "tests"))
    import importlib
    import torch.distributed as dist
    shard = importlib.import_module("7bgzf_amd.shard")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = np.fromfile(path, dtype=np.uint8)
    nb = -(-len(data) // 0xff00)
    lo, hi = shard.block_range(nb, rank, world)
    members = []
    for b in range(lo, hi):
        chunk = bytes(data[b * 0xff00:(b + 1) * 0xff00])
        r, payload = hdtest.oracle_twin(chunk, 1, cap=65536 - 26)
        assert r == 0
        m = np.zeros(65536, dtype=np.uint8)
        p = hdtest.as_u8(payload)
        n = hdtest.oracle().hdo_bgzf_frame(m.ctypes.data, 65536, p.ctypes.data, len(p),
                                           hdtest.oracle_crc32(chunk), len(chunk))
        members.append(bytes(m[:n]))
    span = b"".join(members)
    totals = shard.exchange_totals(len(span))
    bases, total = shard.bases_from_totals(totals)
    with open(out_path, "r+b") as f:          # every rank pwrite()s its span at its base
        f.seek(bases[rank])
        f.write(span)
    dist.barrier()
    if rank == 0:
        assert total == os.path.getsize(out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_block_range_shard_and_offsets(world):
    import torch.multiprocessing as mp
    data = hdtest.synth().fastq_like(11 * 0xff00 + 4321, seed=5)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "in.bin")
        data.tofile(path)
        # single-rank stream
        single = []
        for b in range(12):
            chunk = bytes(data[b * 0xff00:(b + 1) * 0xff00])
            r, payload = hdtest.oracle_twin(chunk, 1, cap=65536 - 26)
            m = np.zeros(65536, dtype=np.uint8)
            p = hdtest.as_u8(payload)
            n = hdtest.oracle().hdo_bgzf_frame(m.ctypes.data, 65536, p.ctypes.data, len(p),
                                               hdtest.oracle_crc32(chunk), len(chunk))
            single.append(bytes(m[:n]))
        want = b"".join(single)
        out_path = os.path.join(d, "out.bgz")
        with open(out_path, "wb") as f:
            f.truncate(len(want))
        mp.spawn(_worker, args=(world, _free_port(), path, out_path), nprocs=world, join=True)
        got = open(out_path, "rb").read()
        assert got == want
        import gzip
        assert gzip.decompress(got + hdtest.pkg().BGZF_EOF) == bytes(data)


def test_block_range_properties():
    import importlib
    shard = importlib.import_module("7bgzf_amd.shard")
    for nb in (0, 1, 7, 8, 263173, 4210753):
        for world in (1, 2, 4, 8):
            spans = [shard.block_range(nb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard.bases_from_totals([5, 0, 7]) == ([0, 5, 5], 12)
