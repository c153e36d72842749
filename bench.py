#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config.

Default workload (BASELINE.json configs[1], the config the metric is quoted on):
  BGZF encode, 0xff00-byte blocks, greedy LZ77 + static Huffman (level 1),
  16 GiB synthetic FASTQ-like bytes resident in HBM, 1 x MI355X.
One "step" = one pass of the hot path over that batch: the encode kernel over
all 263,173 blocks, the size prefix scan (plus, with N > 1, the one RCCL
all_gather of per-rank totals -- SURVEY.md 8(e)) and the gather of the members
into one contiguous BGZF stream.  value = input bytes of ALL ranks / time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode encode|decode]
                  [--level L] [--gib G] [--no-cpu]

N > 1 is launched by the driver with torch.distributed.run, one rank per GPU
(weak scaling: every rank holds its own G GiB shard = a contiguous block range).

The JSON line also carries
  roofline      the dominant kernel against the HBM roofline, timed with events on
                the launch stream inside the timed region (DESIGN.md "Measurement")
  cpu_baseline  the REAL reference per-block path (libdeflate 1.23 through
                libdeflate_deflate, lib/zlibutil.c:179, from oracle/_ref/libref.so)
                on this box's host cores, on a bounded sample of the same workload.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
BLOCK = 0xff00                 # applet/7bgzf.c:146-147 (overridden by --block-kib for MiGz)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", default="encode", choices=["encode", "decode"])
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--gib", type=float, default=16.0, help="input GiB per GPU")
    ap.add_argument("--tile-mib", type=int, default=64, help="host-generated tile replicated on the device")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--data", default="fastq", choices=["fastq", "text", "random"],
                    help="fastq = configs[1..3]; text = config 5 (enwik-like); random = config 1 stand-in")
    ap.add_argument("--slot", type=int, default=0, help="experiment: output slot bytes per block (multiple of 16; 0 = default)")
    ap.add_argument("--block-kib", type=int, default=0, help="0 = BGZF 0xff00-byte blocks; else MiGz blocks of N KiB")
    ap.add_argument("--stream", default="own", choices=["own", "libdeflate6", "zlib6", "libdeflate1"],
                    help="decode mode: who compressed the stream (reference encoders need oracle/_ref/libref.so)")
    return ap.parse_args()


# ---- CPU baseline: the reference's own per-block function on host cores ------------


def cpu_baseline(tile, level, mode, sample_budget_s=16.0):
    so = os.path.join(ROOT, "oracle", "_ref", "libref.so")
    ncores = len(os.sched_getaffinity(0))
    blocks = [tile[i:i + BLOCK] for i in range(0, len(tile) - BLOCK + 1, BLOCK)]
    if os.path.exists(so):
        ref = ctypes.CDLL(so)
        kind = "reference"
        enc, dec = ref.libdeflate_deflate, ref.libdeflate_inflate
        what = "libdeflate 1.23 via libdeflate_%s (lib/zlibutil.c), one call per 0xff00 block" % (
            "deflate" if mode == "encode" else "inflate")
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hdtest
        o = hdtest.oracle()
        kind = "port"
        enc = o.hdo_deflate_twin
        dec = lambda d, dl, s, sl: o.hdo_inflate(d, dl, s, sl, None)
        what = "oracle CPU twin (oracle/_ref not built)"
    vp = ctypes.c_void_p

    def enc_block(blk, out):
        n = ctypes.c_size_t(len(out))
        r = enc(out.ctypes.data_as(vp), ctypes.byref(n), blk.ctypes.data_as(vp), ctypes.c_size_t(len(blk)), level)
        assert r == 0
        return n.value

    comp = None
    if mode == "decode":
        comp = []
        out = np.zeros(BLOCK * 2, dtype=np.uint8)
        for blk in blocks[:256]:
            n = enc_block(blk, out)
            comp.append(out[:n].copy())

    def work(idx, n_iter, res):
        out = np.zeros(BLOCK * 2, dtype=np.uint8)
        done = 0
        t0 = time.perf_counter()
        for k in range(n_iter):
            if mode == "encode":
                blk = blocks[(idx * 7919 + k) % len(blocks)]
                enc_block(blk, out)
            else:
                z = comp[(idx * 31 + k) % len(comp)]
                n = ctypes.c_size_t(BLOCK)
                r = dec(out.ctypes.data_as(vp), ctypes.byref(n), z.ctypes.data_as(vp), ctypes.c_size_t(len(z)))
                assert r == 0 and n.value == BLOCK
            done += BLOCK
        res[idx] = (done, time.perf_counter() - t0)

    # calibrate on one core, then size the all-core run to the CPU-work budget
    res = [None]
    work(0, 64, res)
    per_block = res[0][1] / 64
    one_core = BLOCK / per_block / 1e9
    threads = min(ncores, 16)          # a one-GPU box owns 16 host cores
    n_iter = max(16, int(sample_budget_s / per_block / threads))
    res = [None] * threads
    th = [threading.Thread(target=work, args=(i, n_iter, res)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    return {"value": round(total / wall / 1e9, 4), "unit": "GB/s", "cores": threads, "kind": kind,
            "sample": "%s; %d blocks (%.2f GB) of the same FASTQ-like workload over %d threads, %.1f s CPU work; "
                      "1 core alone: %.4f GB/s" % (what, threads * n_iter, total / 1e9, threads, wall * threads,
                                                   one_core)}


def main():
    global BLOCK
    args = parse()
    if args.block_kib:
        BLOCK = args.block_kib * 1024
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    force_dist = world == 1 and os.environ.get("HD_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    pkg = importlib.import_module("7bgzf_amd")
    dev = importlib.import_module("7bgzf_amd.device")
    synth = importlib.import_module("7bgzf_amd.synth")
    pkg.lib().hipdeflate_init(local)
    if not pkg.available():
        raise SystemExit("no usable MI355X; there is no CPU fallback to measure")

    # ---- synthetic input: a seeded FASTQ-like tile replicated to G GiB in HBM ---------
    tile_bytes = args.tile_mib << 20
    if args.mode == "decode" and args.stream != "own":
        tile_bytes = tile_bytes // BLOCK * BLOCK      # whole blocks per tile: the compressed tile repeats too
    if args.data == "fastq":
        tile_np = synth.fastq_like(tile_bytes, seed=1234 + rank, first_record=1 + rank * 10_000_000)
    elif args.data == "text":
        tile_np = synth.text_like(tile_bytes, seed=4321 + rank)
    else:
        tile_np = synth.random_bytes(tile_bytes, seed=99 + rank)
    assert len(tile_np) == tile_bytes
    total = int(args.gib * (1 << 30))
    reps = max(1, total // tile_bytes)
    total = reps * tile_bytes
    tile = torch.from_numpy(tile_np).cuda()
    data = tile.repeat(reps)
    del tile
    assert data.numel() == total
    off, ln = dev.block_table(total, BLOCK)
    nb = off.numel()

    frame = pkg.FRAME_MIGZ if args.block_kib else pkg.FRAME_BGZF
    hdr = 20 if args.block_kib else 18
    slot = args.slot or (65536 if not args.block_kib else int(pkg.lib().hipdeflate_bound(BLOCK, 9)))
    enc = dev.DeviceDeflate(nb, slot=slot)
    if args.mode == "encode":
        level = args.level
        packed = torch.empty(int(total * (0.75 if (level >= 1 and args.data != "random") else 1.01)) + (1 << 20),
                             dtype=torch.uint8, device="cuda")
    else:
        level = args.level
        if args.stream == "own":
            # the stream to inflate is produced once, untimed, by our own encoder at --level
            # (valid RFC 1951 multi-member BGZF); it then stays resident in HBM
            enc.run(data, off, ln, level=level, frame=frame)
            enc.scan()
            torch.cuda.synchronize()
            comp_total = int(enc.total.item())
            packed = torch.empty(comp_total + 16, dtype=torch.uint8, device="cuda")
            enc.compact(packed)
            in_off = enc.dst_off + hdr
            in_len = (enc.out_len - hdr).to(torch.int32)
            want_crc = enc.crc
        else:
            # BASELINE config 3: the REFERENCE's encoder (libdeflate 1.23 / zlib 1.3.1 built from
            # the reference tree) compresses one tile of whole blocks on the host, untimed; the
            # compressed tile is replicated like the data
            import zlib as _z
            from concurrent.futures import ThreadPoolExecutor
            ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref.so"))
            fn, lvl = {"libdeflate6": (ref.libdeflate_deflate, 6), "libdeflate1": (ref.libdeflate_deflate, 1),
                       "zlib6": (ref.zlib_deflate, 6)}[args.stream]
            tb = tile_np.tobytes()
            nbt = tile_bytes // BLOCK

            def comp(i):
                src = tb[i * BLOCK:(i + 1) * BLOCK]
                dst = ctypes.create_string_buffer(BLOCK * 2)
                n = ctypes.c_size_t(BLOCK * 2)
                assert fn(dst, ctypes.byref(n), src, ctypes.c_size_t(BLOCK), lvl) == 0
                return dst.raw[:n.value], _z.crc32(src)
            with ThreadPoolExecutor(16) as ex:
                res = list(ex.map(comp, range(nbt)))
            lens_t = np.array([len(r[0]) for r in res], dtype=np.int64)
            offs_t = np.concatenate([[0], np.cumsum(lens_t)[:-1]])
            ctile = np.frombuffer(b"".join(r[0] for r in res), dtype=np.uint8)
            packed = torch.from_numpy(ctile.copy()).cuda().repeat(reps)
            comp_total = packed.numel()
            in_off = (torch.from_numpy(offs_t).cuda()[None, :] +
                      (torch.arange(reps, device="cuda", dtype=torch.int64) * len(ctile))[:, None]).reshape(-1)
            in_len = torch.from_numpy(lens_t.astype(np.int32)).cuda().repeat(reps)
            want_crc = torch.from_numpy(np.array([r[1] for r in res], dtype=np.uint32).view(np.int32)).cuda().repeat(reps)
            level = lvl
        out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
        crc = torch.zeros(nb, dtype=torch.int32, device="cuda")
        st = torch.zeros(nb, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        del enc.slots
        out = torch.empty_like(data)

    totals = torch.zeros(world, dtype=torch.int64, device="cuda")
    kern_ms = []

    def step(timed):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if args.mode == "encode":
            e0.record()
            enc.run(data, off, ln, level=level, frame=frame)
            e1.record()
            enc.scan()
            if world > 1 or force_dist:
                # the ONE exchange of the path: per-rank compressed totals -> base offsets
                dist.all_gather_into_tensor(totals, enc.total)
            enc.compact(packed)
        else:
            e0.record()
            dev.device_inflate(packed, in_off, in_len, out, off, ln, out_len, crc, st)
            e1.record()
        if timed:
            kern_ms.append((e0, e1))

    def fence():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1 or force_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- sanity: nothing failed, sizes plausible (parity itself is tests/ -m gpu) -----
    if args.mode == "encode":
        assert int(enc.status.abs().sum()) == 0
        comp_total = int(enc.total.item())
    else:
        assert int(st.abs().sum()) == 0 and torch.equal(out_len, ln)
        assert torch.equal(out[: 4 * BLOCK], data[: 4 * BLOCK]) and torch.equal(out[-BLOCK:], data[-BLOCK:])
        assert torch.equal(crc, want_crc)
    ratio = comp_total / total

    kms = [a.elapsed_time(b) for a, b in kern_ms]
    k_avg_s = sum(kms) / len(kms) / 1e3
    # algorithmic bytes of the dominant kernel per launch:
    #   encode N_in + N_out + 8 B/block (len, crc); decode N_cmp + N_out
    algo_bytes = total + comp_total + 8 * nb
    achieved = algo_bytes / k_avg_s / 1e9
    traffic = None
    tp = os.path.join(ROOT, "profiles", "traffic_%s_l%d.json" % (args.mode, level))
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            if tj.get("input_bytes") == total:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    gbs = total * world * args.steps / elapsed / 1e9

    if rank == 0:
        line = {
            "metric": ("GB/s input compressed, %s blocks" if args.mode == "encode"
                       else "GB/s output produced, %s blocks (inflate)") % (
                           "BGZF 64KiB" if not args.block_kib else "MiGz %d KiB" % args.block_kib),
            "value": round(gbs, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s %s, %s blocks, %s, %.2f GiB %s per GPU (seeded generator, "
                                   "%d MiB tile x %d), %d blocks/GPU, HBM-resident" % (
                                       "MiGz" if args.block_kib else "BGZF", args.mode,
                                       ("%d KiB" % args.block_kib) if args.block_kib else "0xff00-byte",
                                       ("stream from the reference's %s (built from the reference tree)" % args.stream)
                                       if (args.mode == "decode" and args.stream != "own") else
                                       ("level %d: greedy LZ77 + static Huffman" % level) if level == 1 else
                                       ("level %d: %s LZ77 + dynamic Huffman" % (level, "lazy" if level >= 5 else "greedy")) if level >= 2
                                       else "level 0: stored",
                                       total / 2 ** 30, {"fastq": "FASTQ-like", "text": "enwik-like text", "random": "random bytes"}[args.data],
                                       args.tile_mib, reps, nb),
                       "blocks_per_gpu": nb, "ratio": round(ratio, 4), "parallelism": "block-range shard x%d" % world,
                       "step": "encode kernel + size scan (+ all_gather of totals) + compact" if args.mode == "encode"
                       else "inflate kernel", "stream": args.stream if args.mode == "decode" else None},
            "roofline": {"bound": "hbm", "kernel": "k_deflate_static" if args.mode == "encode" and level <= 1
                         else ("k_deflate_dynamic" if args.mode == "encode" else "k_inflate"),
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms_avg": round(k_avg_s * 1e3, 3)},
        }
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(tile_np, max(level, 1), args.mode)
        elif not args.no_cpu:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
