#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config.

Default workload (BASELINE.json configs[1], the config the metric is quoted on):
  BGZF encode, 0xff00-byte blocks, greedy LZ77 + static Huffman (level 1),
  16 GiB synthetic FASTQ-like bytes resident in HBM, 1 x MI355X.
One "step" = one pass of the hot path over that batch: the encode kernel over
all 263,173 blocks, the size prefix scan and the gather of the members into one
contiguous BGZF stream.  value = input bytes of ALL ranks / time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode encode|decode]
                  [--level L] [--gib G] [--no-cpu] [--no-extra]

N > 1: one rank per GPU.  The driver launches the ranks itself (torch.distributed.run, WORLD_SIZE
in the environment); called plainly as `python bench.py --gpus N`, bench.py starts that launcher
as a CHILD process before anything here touches torch or HIP, passes its one JSON line through
and exits with its code.  WORLD_SIZE != --gpus is an error, never a silent one-GPU run.  Weak
scaling: every rank holds its own G GiB shard = a contiguous block range of a G*N GiB stream;
G defaults to 16 at N = 1 (BASELINE config 2) and to 32 at N > 1 (config 4: 256 GiB over 8 GPUs).  The step then contains the ONE exchange of the path (SURVEY.md
8(e)): all_gather of the per-rank compressed totals (7bgzf_amd/shard.py, RCCL) ->
this rank's base -> member offsets in the whole stream (scan with that base) ->
gather into the rank's span (what the rank would pwrite() at `base`).
HD_BENCH_FORCE_DIST=1 runs that path at world size 1.

The JSON line also carries
  roofline      the dominant kernel against the HBM roofline, timed with events on
                the launch stream inside the timed region (DESIGN.md "Measurement")
  cpu_baseline  the REAL reference per-block path (libdeflate 1.23 through
                libdeflate_deflate, lib/zlibutil.c:179, from oracle/_ref/libref.so)
                on this box's host cores, on a bounded sample of the same workload;
                plus the reused-compressor figure, one core alone, and our CPU twin.
  configs       (N = 1, default invocation only) the other GPU configs of BASELINE.json
                measured in the same run: level 2 on the same data, decode of the
                reference's libdeflate-6 stream (config 3), MiGz 1 MiB level 6 on
                enwik-like text (config 5).
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
BGZF_BLOCK = 0xff00            # applet/7bgzf.c:146-147


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", default="encode", choices=["encode", "decode"])
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--gib", type=float, default=None,
                    help="input GiB per GPU (default: 16 at N = 1 = config 2; 32 at N > 1 = config 4's 256 GiB over 8 GPUs)")
    ap.add_argument("--tile-mib", type=int, default=64, help="host-generated tile replicated on the device")
    ap.add_argument("--first-record", type=int, default=0,
                    help="FASTQ-like data: the first record id (default: nine-digit ids, offset per tile; 1 = rounds 1-4's data, identical tiles)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="only the headline config (profiling runs)")
    ap.add_argument("--data", default="fastq", choices=["fastq", "text", "random"],
                    help="fastq = configs[1..3]; text = config 5 (enwik-like); random = config 1 stand-in")
    ap.add_argument("--slot", type=int, default=0, help="experiment: output slot bytes per block (multiple of 16; 0 = default)")
    ap.add_argument("--block-kib", type=int, default=0, help="0 = BGZF 0xff00-byte blocks; else MiGz blocks of N KiB")
    ap.add_argument("--stream", default="own", choices=["own", "libdeflate6", "zlib6", "libdeflate1"],
                    help="decode mode: who compressed the stream (reference encoders need oracle/_ref/libref.so)")
    args = ap.parse_args()
    args.gib_default = args.gib is None
    if args.gib is None:
        args.gib = 16.0 if args.gpus <= 1 else 32.0
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start N ranks as a CHILD process
    (python -m torch.distributed.run, rendezvous on 127.0.0.1) before this process has imported torch
    or made any HIP call, hand its JSON line through, return its exit code.  Never an exec."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("HD_BENCH_LAUNCH", None)
    if args.gpus == 1:
        env["HD_BENCH_FORCE_DIST"] = "1"         # the launcher path at N = 1 runs the N > 1 code
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for ln in p.stdout:
        if ln.startswith("{"):
            lines.append(ln)
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if rc == 0 and len(lines) != 1:
        sys.stderr.write("bench.py: expected ONE JSON line from rank 0, got %d\n" % len(lines))
        rc = 1
    for ln in lines:
        sys.stdout.write(ln)
    sys.stdout.flush()
    return rc


# ---- CPU baseline: the reference's own per-block function on host cores ------------


def _pool(work, threads, n_iter):
    """run work(idx, n_iter) -> bytes done on `threads` python threads (ctypes drops the GIL); GB/s, wall s"""
    res = [0] * threads

    def run(i):
        res[i] = work(i, n_iter)
    th = [threading.Thread(target=run, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    return sum(res) / wall / 1e9, wall


def cpu_baseline(tile, level, mode, block, sample_budget_s=12.0):
    so = os.path.join(ROOT, "oracle", "_ref", "libref.so")
    ncores = len(os.sched_getaffinity(0))
    threads = min(ncores, 16)          # a one-GPU box owns 16 host cores
    blocks = [tile[i:i + block] for i in range(0, len(tile) - block + 1, block)][:1024]
    vp = ctypes.c_void_p
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hdtest
    o = hdtest.oracle()
    have_ref = os.path.exists(so)
    if have_ref:
        ref = ctypes.CDLL(so)
        kind = "reference"
        enc, dec = ref.libdeflate_deflate, ref.libdeflate_inflate
        ref.libdeflate_alloc_compressor.restype = vp
        ref.libdeflate_alloc_compressor.argtypes = [ctypes.c_int]
        ref.libdeflate_free_compressor.argtypes = [vp]
        ref.libdeflate_deflate_compress.restype = ctypes.c_size_t
        ref.libdeflate_deflate_compress.argtypes = [vp, vp, ctypes.c_size_t, vp, ctypes.c_size_t]
        what = "libdeflate 1.23 via libdeflate_%s (lib/zlibutil.c), one call per %d-byte block" % (
            "deflate" if mode == "encode" else "inflate", block)
    else:
        kind = "port"
        enc = o.hdo_deflate_twin
        dec = lambda d, dl, s, sl: o.hdo_inflate(d, dl, s, sl, None)
        what = "oracle CPU twin (oracle/_ref not built)"

    def enc_block(fn, blk, out):
        n = ctypes.c_size_t(len(out))
        r = fn(out.ctypes.data_as(vp), ctypes.byref(n), blk.ctypes.data_as(vp), ctypes.c_size_t(len(blk)), level)
        assert r == 0
        return n.value

    comp = None
    if mode == "decode":
        comp = []
        out = np.zeros(block * 2, dtype=np.uint8)
        for blk in blocks[:256]:
            n = enc_block(enc, blk, out)
            comp.append(out[:n].copy())

    def adapter_work(fn):
        def work(idx, n_iter):
            out = np.zeros(block * 2, dtype=np.uint8)
            done = 0
            for k in range(n_iter):
                if mode == "encode":
                    enc_block(fn, blocks[(idx * 7919 + k) % len(blocks)], out)
                else:
                    z = comp[(idx * 31 + k) % len(comp)]
                    n = ctypes.c_size_t(block)
                    r = dec(out.ctypes.data_as(vp), ctypes.byref(n), z.ctypes.data_as(vp), ctypes.c_size_t(len(z)))
                    assert r == 0 and n.value == block
                done += block
            return done
        return work

    def reused_work(idx, n_iter):
        # one libdeflate compressor per thread, reused: what the reference would
        # cost without the alloc/free per call of lib/zlibutil.c:186-188
        c = ref.libdeflate_alloc_compressor(level)
        out = np.zeros(block * 2, dtype=np.uint8)
        done = 0
        for k in range(n_iter):
            blk = blocks[(idx * 7919 + k) % len(blocks)]
            assert ref.libdeflate_deflate_compress(c, blk.ctypes.data_as(vp), len(blk), out.ctypes.data_as(vp), len(out))
            done += block
        ref.libdeflate_free_compressor(c)
        return done

    def measure(work, share):
        """calibrate on one core, then one-core and all-core figures inside `share` of the budget"""
        t0 = time.perf_counter()
        work(0, 16)
        per_block = (time.perf_counter() - t0) / 16
        budget = sample_budget_s * share
        n1 = max(16, int(budget * 0.25 / per_block))
        one, _ = _pool(work, 1, n1)
        n_all = max(16, int(budget * 0.75 / per_block / threads))
        allc, wall = _pool(work, threads, n_all)
        return one, allc, wall, n_all

    out = {}
    # The reference function on real pthreads (tools/hook_bench.c linked against oracle/_ref/libref.so) is the reported
    # figure where the harness exists: Python threads hand the GIL around between the ctypes calls and sell the reference
    # short (round 2: by 2x), so they are only the fallback.  The reference's LD_PRELOAD hook (bgzf_compress.c: codec +
    # CRC-32 + framing per call) is timed the same way.
    harness = os.path.join(ROOT, "oracle", "_ref", "hook_bench_ref")
    done = False
    if have_ref and mode == "encode" and os.path.exists(harness):
        import subprocess
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".bin") as tf:
            tile[: min(len(tile), 64 << 20)].tofile(tf.name)

            def run(nthr, codec, env_method=None):
                env = dict(os.environ)
                if env_method:
                    env["BGZF_METHOD"] = env_method
                cmd = [harness, tf.name, str(nthr), "2", str(block)] + ([codec] if codec else [])
                p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=120)
                return json.loads(p.stdout.strip().split("\n")[-1])["GBps_in"] if p.returncode == 0 else None
            try:
                # the comparator is libdeflate with ONE REUSED compressor per thread (the reference's
                # adapter allocates one per call, lib/zlibutil.c:186-188; that figure is kept beside it)
                r_all = run(threads, "libdeflate_reused:%d" % level)
                r_one = run(1, "libdeflate_reused:%d" % level)
                c_all = run(threads, "libdeflate_deflate:%d" % level)
                c_one = run(1, "libdeflate_deflate:%d" % level)
                hook = run(threads, None, "libdeflate%d" % level) if block <= 0xff00 else None
                if r_all:
                    out.update({"value": round(r_all, 4), "unit": "GB/s", "cores": threads, "kind": kind,
                                "one_core": round(r_one or 0, 4),
                                "sample": "libdeflate 1.23 libdeflate_deflate_compress level %d, one reused compressor per "
                                          "thread, one call per %d-byte block of the same workload from %d pthreads for 2 s = "
                                          "%d core-seconds (tools/hook_bench.c against oracle/_ref/libref.so)"
                                          % (level, block, threads, 2 * threads)})
                    out["reused_compressor"] = {"one_core": round(r_one or 0, 4), "all_cores": round(r_all, 4),
                                                "what": "the same figure (cpu_baseline.value), kept under its old key"}
                    if c_all:
                        out["per_call_adapter"] = {"all_cores": round(c_all, 4), "one_core": round(c_one or 0, 4),
                                                   "what": "libdeflate_deflate (lib/zlibutil.c:179-192): compressor allocated "
                                                           "and freed in every call, as the reference's applets run it"}
                    done = True
                if hook:
                    out["reference_hook"] = {"all_cores": round(hook, 4), "what": "bgzf_compress (bgzf_compress.c:39) with "
                                             "BGZF_METHOD=libdeflate%d from %d pthreads: codec + CRC-32 + framing per call" % (level, threads)}
            except Exception as ex:
                out["c_harness_error"] = "%s: %s" % (type(ex).__name__, ex)
    if not done:
        one, allc, wall, n_all = measure(adapter_work(enc), 0.45 if mode == "encode" else 1.0)
        out.update({"value": round(allc, 4), "unit": "GB/s", "cores": threads, "kind": kind,
                    "one_core": round(one, 4),
                    "sample": "%s; %d blocks (%.2f GB) of the same workload over %d python threads, %.1f s CPU work"
                              % (what, threads * n_all, threads * n_all * block / 1e9, threads, wall * threads)})
    if mode == "encode":
        if have_ref and "reused_compressor" not in out:
            one, allc, _, _ = measure(reused_work, 0.15)
            out["reused_compressor"] = {"one_core": round(one, 4), "all_cores": round(allc, 4),
                                        "what": "libdeflate_deflate_compress level %d, one compressor per thread" % level}
        one, allc, _, _ = measure(adapter_work(o.hdo_deflate_twin), 0.15)
        out["twin"] = {"one_core": round(one, 4), "all_cores": round(allc, 4),
                       "what": "oracle/hd_deflate_twin.c level %d: the serial restatement of the HIP encoder "
                               "(same bytes as the kernel)" % level}
    return out


# ---- the GPU side -------------------------------------------------------------------


class Bench:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        if self.world != max(args.gpus, 1):
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or call bench.py "
                             "without a launcher and let it start the ranks)" % (args.gpus, self.world))
        # HD_BENCH_DEVICE: every rank on that one card (the two-rank rehearsal on a one-GPU box, with
        # HD_BENCH_DIST_BACKEND=gloo: RCCL wants one device per rank)
        local = int(os.environ.get("HD_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(local)
        force = self.world == 1 and os.environ.get("HD_BENCH_FORCE_DIST") == "1"
        self.use_dist = self.world > 1 or force
        self.backend = os.environ.get("HD_BENCH_DIST_BACKEND", "nccl")
        self.coll_dev = "cuda" if self.backend == "nccl" else "cpu"
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        self.pkg = importlib.import_module("7bgzf_amd")
        self.dev = importlib.import_module("7bgzf_amd.device")
        self.synth = importlib.import_module("7bgzf_amd.synth")
        self.shard = importlib.import_module("7bgzf_amd.shard")
        self.pkg.lib().hipdeflate_init(local)
        if not self.pkg.available():
            raise SystemExit("no usable MI355X; there is no CPU fallback to measure")

    # a seeded tile replicated to G GiB in HBM.  FASTQ-like input of the ENCODE runs: every tile carries its own record ids
    # ("tile with per-tile record-id offset") -- the ids are nine digits wide from the first record on, so a
    # tile differs from tile 0 in the digits of its two id fields per record and in nothing else, and the digits are written
    # on the device (no second pass of the CPU generator).  The decode runs on the reference's streams replicate one
    # compressed tile (the reference's encoder runs on the host, once) and say so.
    def make_data(self, kind, block, whole_blocks):
        torch, args = self.torch, self.args
        tile_bytes = args.tile_mib << 20
        if whole_blocks:
            tile_bytes = tile_bytes // block * block      # whole blocks per tile: a compressed tile repeats too
        r = self.rank
        first = 100_000_000 * (1 + r)                     # nine digits up to 999,999,999: 8 ranks x 512 tiles x ~205 k records fit
        if args.first_record:
            first = args.first_record                     # (--first-record 1: rounds 1-4's data -- ids from 1, identical tiles)
        if kind == "fastq":
            tile_np = self.synth.fastq_like(tile_bytes, seed=1234 + r, first_record=first)
        elif kind == "text":
            tile_np = self.synth.text_like(tile_bytes, seed=4321 + r)
        else:
            tile_np = self.synth.random_bytes(tile_bytes, seed=99 + r)
        assert len(tile_np) == tile_bytes
        total = int(args.gib * (1 << 30))
        reps = max(1, total // tile_bytes)
        data = torch.from_numpy(tile_np).cuda().repeat(reps)
        self.tiles_vary = False
        if kind == "fastq" and not whole_blocks and reps > 1:
            import re
            tb = tile_np.tobytes()
            pos = np.array([m.start(1) for m in re.finditer(rb"@SRR000001\.(\d{9}) \d{9}/1\n", tb)], dtype=np.int64)
            nrec = len(pos)
            if nrec == 0 and args.first_record:
                return tile_np, data, reps                   # ids not nine digits wide: identical tiles, as rounds 1-4 had them
            assert nrec > tile_bytes // 400 and tb[pos[0]:pos[0] + 9] == b"%d" % first
            dpos = torch.from_numpy(pos).cuda()
            ids0 = torch.arange(nrec, device="cuda", dtype=torch.int64) + first
            pw = torch.tensor([10 ** (8 - d) for d in range(9)], device="cuda", dtype=torch.int64)
            col = torch.arange(9, device="cuda", dtype=torch.int64)
            for t in range(1, reps):
                ids = ids0 + t * (nrec + 1)                      # (+ 1: the tile's last, cut record)
                assert int(ids[-1]) < 1_000_000_000
                dig = ((ids[:, None] // pw[None, :]) % 10 + 48).to(torch.uint8)
                at = t * tile_bytes + dpos[:, None] + col[None, :]
                data[at] = dig
                data[at + 10] = dig                              # the second id field, behind the blank
            self.tiles_vary = True
            del dpos, ids0
        return tile_np, data, reps

    def fence(self):
        self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, step, steps, warmup):
        torch = self.torch
        drain = getattr(step, "drain", lambda: None)     # a software-pipelined step finishes its last batch here
        for _ in range(warmup):
            step(None)
        drain()
        self.fence()
        evs = []
        t0 = time.perf_counter()
        for _ in range(steps):
            step(evs)
        drain()
        self.fence()
        elapsed = time.perf_counter() - t0
        self.rank_elapsed = [elapsed]
        if self.use_dist:
            # every rank's own time (the line carries them, so a reader sees how many ranks the collective saw);
            # the job's time is the slowest rank's
            self.rank_elapsed = self.gather_f64(elapsed)
            elapsed = max(self.rank_elapsed)
        kms = [a.elapsed_time(b) for a, b in evs]
        return elapsed, sum(kms) / len(kms) / 1e3

    def gather_f64(self, x):
        torch = self.torch
        mine = torch.tensor([float(x)], dtype=torch.float64, device=self.coll_dev)
        allt = torch.zeros(self.world, dtype=torch.float64, device=self.coll_dev)
        self.dist.all_gather_into_tensor(allt, mine)
        return [float(v) for v in allt.cpu()]

    def encode(self, data, block, level, migz, steps, warmup, slot_arg=0, incompressible=False):
        """-> dict(elapsed, k_avg_s, total, comp_total, nb, ...) for `steps` passes of the encode path"""
        torch, pkg, dev = self.torch, self.pkg, self.dev
        total = data.numel()
        off, ln = dev.block_table(total, block)
        nb = off.numel()
        frame = pkg.FRAME_MIGZ if migz else pkg.FRAME_BGZF
        slot = slot_arg or (65536 if not migz else int(pkg.lib().hipdeflate_bound(block, 9)))
        # Two sets of slot / span buffers: the passes are software-pipelined, as a stream of batches is -- the size
        # scan, the exchange of totals (N > 1) and the gather of batch k run on a second stream beside the encode
        # kernel of batch k + 1 (the gather is an HBM copy, the encoder is bound by instruction issue).  Every pass
        # still does all of its work inside the timed region: the last gather is drained before the clock stops.
        # (room for every member stored: a block the encoder gives up is written stored, and a span sized for the expected
        # ratio would then be overrun by the gather -- 288 GB of HBM have the room)
        span_bytes = int(total * 1.01) + (1 << 20)
        encs = [dev.DeviceDeflate(nb, slot=slot), dev.DeviceDeflate(nb, slot=slot)]
        packs = [torch.empty(span_bytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
        main, side = torch.cuda.current_stream(), torch.cuda.Stream()
        coded = [torch.cuda.Event(), torch.cuda.Event()]         # encode of buffer i queued up to here
        gathered = [torch.cuda.Event(), torch.cuda.Event()]      # buffer i's slots are free again
        state = {"k": 0, "pending": None}

        def gather(i):
            enc = encs[i]
            with torch.cuda.stream(side):
                side.wait_event(coded[i])
                enc.scan()                                       # local offsets + this rank's total
                base = 0
                if self.use_dist:
                    # the ONE exchange of the path: per-rank compressed totals -> base offsets
                    totals = self.shard.exchange_totals(int(enc.total.item()), device=self.coll_dev)
                    bases, grand = self.shard.bases_from_totals(totals)
                    base = bases[self.rank]
                    enc.scan(base=base)                          # member offsets in the whole stream
                    state.update(totals=totals, bases=bases, grand=grand)
                enc.compact(packs[i], span_base=base)            # members -> this rank's span
                gathered[i].record(side)
            state["base"] = base
            state["last"] = i

        def step(evs):
            i = state["k"] & 1
            state["k"] += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            main.wait_event(gathered[i])                         # (not yet recorded the first two times: no wait)
            e0.record()
            encs[i].run(data, off, ln, level=level, frame=frame)
            e1.record()
            coded[i].record(main)
            if state["pending"] is not None:
                gather(state["pending"])                         # beside the kernel just queued
            state["pending"] = i
            if evs is not None:
                evs.append((e0, e1))

        def drain():
            if state["pending"] is not None:
                gather(state["pending"])
                state["pending"] = None
            main.wait_stream(side)
        step.drain = drain

        elapsed, k_avg_s = self.timed(step, steps, warmup)
        # ---- sanity: nothing failed, sizes plausible (parity itself is tests/ -m gpu) -----
        enc, packed = encs[state["last"]], packs[state["last"]]
        assert int(encs[0].status.abs().sum()) == 0 and int(encs[1].status.abs().sum()) == 0
        comp_total = int(enc.total.item())
        assert int(enc.dst_off[0].item()) == state["base"]
        assert int(enc.dst_off[-1].item()) + int(enc.out_len[-1].item()) == state["base"] + comp_total
        if self.use_dist:
            assert state["totals"][self.rank] == comp_total
            assert state["grand"] == state["bases"][-1] + state["totals"][-1] == sum(state["totals"])
            assert len(state["totals"]) == self.world
            # what every rank found at the head of its span: its first member's offset in the WHOLE stream
            state["first_member_offset"] = [int(v) for v in self.gather_f64(int(enc.dst_off[0].item()))]
            state["rank_elapsed"] = list(self.rank_elapsed)
        # the span starts with a member header and the last member ends where the span ends
        head = bytes(packed[:4].cpu().numpy())
        assert head == b"\x1f\x8b\x08\x04", head
        res = dict(elapsed=elapsed, k_avg_s=k_avg_s, total=total, comp_total=comp_total, nb=nb, frame=frame,
                   off=off, ln=ln, enc=enc, packed=packed, hdr=20 if migz else 18, dist=dict(state))
        return res

    def verify_encode(self, res, data, block):
        """UNTIMED, after the timed region: the whole packed stream of the last pass is inflated on the device and compared
        with the input byte for byte, every member's CRC-32 with the one its trailer carries (the encoder's own)."""
        torch, dev = self.torch, self.dev
        enc, packed = res["enc"], res["packed"]
        off, ln = res["off"], res["ln"]
        nb = off.numel()
        in_off = enc.dst_off - res["dist"]["base"] + res["hdr"]
        in_len = (enc.out_len - res["hdr"]).to(torch.int32)       # payload + trailer: the inflater stops at BFINAL
        out = torch.empty_like(data)
        out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
        crc = torch.zeros(nb, dtype=torch.int32, device="cuda")
        st = torch.zeros(nb, dtype=torch.int32, device="cuda")
        dev.device_inflate(packed, in_off, in_len, out, off, ln, out_len, crc, st)
        torch.cuda.synchronize()
        ok = (int(st.abs().sum()) == 0 and torch.equal(out_len, ln) and torch.equal(out, data) and torch.equal(crc, enc.crc))
        # the CRC-32 / ISIZE fields of every member's trailer, read back from the packed stream
        end = (enc.dst_off - res["dist"]["base"] + enc.out_len.to(torch.int64))
        idx = (end[:, None] - 8 + torch.arange(8, device="cuda")[None, :]).reshape(-1)
        trl = packed[idx].reshape(nb, 8).to(torch.int64)
        t_crc = trl[:, 0] | (trl[:, 1] << 8) | (trl[:, 2] << 16) | (trl[:, 3] << 24)
        t_isz = trl[:, 4] | (trl[:, 5] << 8) | (trl[:, 6] << 16) | (trl[:, 7] << 24)
        ok = ok and torch.equal(t_crc, enc.crc.to(torch.int64) & 0xffffffff) and torch.equal(t_isz, ln.to(torch.int64))
        del out
        if not ok:
            raise AssertionError("full-size verification failed: the packed stream does not inflate back to the input")
        # the one timing-dependent byte path of the encoders (a workgroup parse that gave a table turn up: the block is then
        # written stored, valid but not the twin's bytes) must not have been taken
        stalls = int(self.pkg.lib().hipdeflate_stall_count())
        if stalls:
            raise AssertionError("%d workgroup-parse stalls: the measured run did not write the twin's bytes" % stalls)
        return {"bytes": int(data.numel()), "members": int(nb), "stalls": stalls,
                "how": "untimed: device inflate of the whole packed stream == input (torch.equal), per-member CRC-32 == "
                       "encoder's == trailer field, ISIZE == block length; hipdeflate_stall_count() == 0"}

    def decode(self, data, packed, in_off, in_len, want_crc, block, steps, warmup):
        torch, dev = self.torch, self.dev
        total = data.numel()
        off, ln = dev.block_table(total, block)
        nb = off.numel()
        out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
        crc = torch.zeros(nb, dtype=torch.int32, device="cuda")
        st = torch.zeros(nb, dtype=torch.int32, device="cuda")
        out = torch.empty_like(data)

        def step(evs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dev.device_inflate(packed, in_off, in_len, out, off, ln, out_len, crc, st)
            e1.record()
            if evs is not None:
                evs.append((e0, e1))

        elapsed, k_avg_s = self.timed(step, steps, warmup)
        assert int(st.abs().sum()) == 0 and torch.equal(out_len, ln)
        assert torch.equal(out[: 4 * block], data[: 4 * block]) and torch.equal(out[-block:], data[-block:])
        assert torch.equal(crc, want_crc)
        return dict(elapsed=elapsed, k_avg_s=k_avg_s, total=total, comp_total=int(packed.numel()), nb=nb)

    def reference_stream(self, tile_np, reps, block, which):
        """BASELINE config 3: the REFERENCE's encoder (libdeflate 1.23 / zlib 1.3.1 built from the reference
        tree) compresses one tile of whole blocks on the host, untimed; the compressed tile is replicated
        like the data"""
        import zlib as _z
        from concurrent.futures import ThreadPoolExecutor
        torch = self.torch
        ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref.so"))
        fn, lvl = {"libdeflate6": (ref.libdeflate_deflate, 6), "libdeflate1": (ref.libdeflate_deflate, 1),
                   "zlib6": (ref.zlib_deflate, 6)}[which]
        tb = tile_np.tobytes()
        nbt = len(tb) // block

        def comp(i):
            src = tb[i * block:(i + 1) * block]
            dst = ctypes.create_string_buffer(block * 2)
            n = ctypes.c_size_t(block * 2)
            assert fn(dst, ctypes.byref(n), src, ctypes.c_size_t(block), lvl) == 0
            return dst.raw[:n.value], _z.crc32(src)
        with ThreadPoolExecutor(16) as ex:
            res = list(ex.map(comp, range(nbt)))
        lens_t = np.array([len(r[0]) for r in res], dtype=np.int64)
        offs_t = np.concatenate([[0], np.cumsum(lens_t)[:-1]])
        ctile = np.frombuffer(b"".join(r[0] for r in res), dtype=np.uint8)
        packed = torch.from_numpy(ctile.copy()).cuda().repeat(reps)
        in_off = (torch.from_numpy(offs_t).cuda()[None, :] +
                  (torch.arange(reps, device="cuda", dtype=torch.int64) * len(ctile))[:, None]).reshape(-1)
        in_len = torch.from_numpy(lens_t.astype(np.int32)).cuda().repeat(reps)
        want_crc = torch.from_numpy(np.array([r[1] for r in res], dtype=np.uint32).view(np.int32)).cuda().repeat(reps)
        return packed, in_off, in_len, want_crc, lvl

    def free(self):
        import gc
        gc.collect()
        self.torch.cuda.empty_cache()


def level_name(level):
    if level <= 0:
        return "level 0: stored"
    if level == 1:
        return "level 1: greedy LZ77 + static Huffman"
    if level == 2:
        return "level 2: greedy LZ77 (one wavefront's 4 KiB window) + dynamic Huffman"
    ways = 4 if level >= 6 else 2 if level == 5 else 1
    return "level %d: %s LZ77 (workgroup-shared 32 KiB window, %d x %d-way table, block splitting) + dynamic Huffman" % (
        level, "lazy" if level >= 4 else "greedy", 32768 // ways, ways)


def summary(res, steps, world=1, mode="encode"):
    """the per-config figures: GB/s of uncompressed bytes, kernel time, roofline fraction"""
    # encode N_in + N_out + 8 B/block (len, crc); decode N_cmp + N_out
    algo = res["total"] + res["comp_total"] + (8 * res["nb"] if mode == "encode" else 0)
    ach = algo / res["k_avg_s"] / 1e9
    return {"value": round(res["total"] * world * steps / res["elapsed"] / 1e9, 3), "unit": "GB/s", "steps": steps,
            "ms_per_step": round(res["elapsed"] / steps * 1e3, 3), "kernel_ms_avg": round(res["k_avg_s"] * 1e3, 3),
            "ratio": round(res["comp_total"] / res["total"], 4), "blocks": res["nb"], "input_bytes": int(res["total"]),
            "algorithmic_bytes_per_launch": algo, "achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5)}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("HD_BENCH_LAUNCH") == "1"):
        sys.exit(launch_ranks(args))             # before torch / HIP are touched in this process
    block = args.block_kib * 1024 if args.block_kib else BGZF_BLOCK
    cpu = None
    if not args.no_cpu and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # The CPU legs come FIRST (a synthetic tile of the same generator, host only): the GPU phase then runs
        # in one piece at the end of the process instead of being followed by ~10 s of host-only work
        synth = importlib.import_module("7bgzf_amd.synth")
        tb = min(args.tile_mib, 64) << 20
        ctile = (synth.fastq_like(tb, seed=1234, first_record=100_000_000) if args.data == "fastq" else
                 synth.text_like(tb, seed=4321) if args.data == "text" else synth.random_bytes(tb, seed=99))
        cpu = cpu_baseline(ctile, max(args.level, 1), args.mode, block)
        del ctile
    B = Bench(args)
    torch = B.torch
    world, rank = B.world, B.rank
    decode_ref = args.mode == "decode" and args.stream != "own"
    tile_np, data, reps = B.make_data(args.data, block, whole_blocks=decode_ref)
    total = data.numel()
    level = args.level
    if args.mode == "encode":
        res = B.encode(data, block, level, bool(args.block_kib), args.steps, args.warmup, slot_arg=args.slot,
                       incompressible=args.data == "random")
    else:
        if args.stream == "own":
            # the stream to inflate is produced once, untimed, by our own encoder at --level
            # (valid RFC 1951 multi-member BGZF); it then stays resident in HBM
            e = B.encode(data, block, level, bool(args.block_kib), 1, 0)
            enc = e["enc"]
            packed = e["packed"][: e["comp_total"] + 16]
            in_off = enc.dst_off - e["dist"]["base"] + e["hdr"]
            in_len = (enc.out_len - e["hdr"]).to(torch.int32)
            want_crc = enc.crc
            del enc.slots
        else:
            packed, in_off, in_len, want_crc, level = B.reference_stream(tile_np, reps, block, args.stream)
        B.free()
        res = B.decode(data, packed, in_off, in_len, want_crc, block, args.steps, args.warmup)
    s = summary(res, args.steps, world, args.mode)
    verified = B.verify_encode(res, data, block) if args.mode == "encode" else {
        "how": "every pass: status, lengths, per-block CRC-32 against the reference stream's; first and last blocks byte for byte"}

    # HBM bytes per launch from the PMC counters: NOT measured in this run (counters need rocprofv3 around the
    # process); the tracked file holds the figure of the same command under `rocprofv3 --pmc` (tools/prof_round.sh)
    traffic, traffic_source = None, None
    tp = os.path.join(ROOT, "profiles", "traffic_%s_l%d.json" % (args.mode, level))
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            if tj.get("input_bytes") == total:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and WRITE_SIZE passes over this "
                                  "command in an earlier run; not measured by this process" % os.path.basename(tp))
        except Exception:
            traffic = None

    line = None
    if rank == 0:
        shard_txt = "HBM-resident" if world == 1 else \
            "HBM-resident; each rank = its own %.2f GiB shard (a contiguous block range) of a %.0f GiB stream%s" % (
                total / 2 ** 30, total * world / 2 ** 30,
                " (BASELINE config 4: 256 GiB over 8 GPUs = 32 GiB per rank, kept per rank at every N > 1)"
                if args.gib_default else "")
        line = {
            "metric": ("GB/s input compressed, %s blocks" if args.mode == "encode"
                       else "GB/s output produced, %s blocks (inflate)") % (
                           "BGZF 64KiB" if not args.block_kib else "MiGz %d KiB" % args.block_kib),
            "value": s["value"], "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": s["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s %s, %s blocks, %s, %.2f GiB %s per GPU (seeded generator, "
                                   "%d MiB tile x %d%s), %d blocks/GPU, %s" % (
                                       "MiGz" if args.block_kib else "BGZF", args.mode,
                                       ("%d KiB" % args.block_kib) if args.block_kib else "0xff00-byte",
                                       ("stream from the reference's %s (built from the reference tree)" % args.stream)
                                       if decode_ref else level_name(level),
                                       total / 2 ** 30, {"fastq": "FASTQ-like", "text": "enwik-like text",
                                                         "random": "random bytes"}[args.data],
                                       args.tile_mib, reps, ", record ids offset per tile" if B.tiles_vary else
                                       (", identical tiles" if reps > 1 else ""), res["nb"], shard_txt),
                       "blocks_per_gpu": res["nb"], "input_bytes_per_gpu": int(total), "ratio": s["ratio"], "parallelism": "block-range shard x%d" % world,
                       "step": ("encode kernel + size scan + %sgather into the contiguous stream; passes are "
                                "software-pipelined as a stream of batches is: scan and gather of pass k run on a second "
                                "HIP stream beside the encode kernel of pass k + 1 (two sets of buffers), the last gather "
                                "is drained inside the timed region" % (
                           "all_gather of per-rank totals (RCCL) + scan with this rank's base + " if B.use_dist else ""))
                       if args.mode == "encode" else "inflate kernel",
                       "stream": args.stream if args.mode == "decode" else None},
            "roofline": {"bound": "hbm", "kernel": "k_deflate_static" if args.mode == "encode" and level <= 1
                         else ("k_parse_wg (parse) + k_deflate_dynamic<EMIT>" if args.mode == "encode" and level >= 3 else
                               "k_deflate_static<TOK> (parse) + k_deflate_dynamic<EMIT>" if args.mode == "encode" else "k_inflate"),
                         "achieved": s["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": s["frac"], "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": s["algorithmic_bytes_per_launch"],
                         "kernel_ms_avg": s["kernel_ms_avg"]},
        }
        line["verified"] = verified
        line["ranks_seen"] = len(B.rank_elapsed)
        line["value_per_rank"] = [round(total * args.steps / e / 1e9, 3) for e in B.rank_elapsed]
        if B.use_dist and args.mode == "encode":
            d = res["dist"]
            line["ranks_seen"] = len(d["totals"])        # length of the all-gathered totals: what the collective saw
            line["config"]["stream_offsets"] = {"totals": d["totals"], "bases": d["bases"], "stream_bytes": d["grand"],
                                                "first_member_offset": d["first_member_offset"]}
            line["config"]["collective"] = "all_gather of one int64 per rank, backend %s" % B.backend
            try:
                line["config"]["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if B.backend == "nccl" else None
            except Exception:
                line["config"]["rccl_version"] = None

    # ---- the other GPU configs of BASELINE.json, same run (default invocation, one GPU) ----------
    default_run = (args.mode == "encode" and level == 1 and args.data == "fastq" and not args.block_kib
                   and world == 1 and not B.use_dist and not args.no_extra and not args.slot and args.gib == 16.0)
    if default_run:
        xs, xw = min(args.steps, 5), 2
        configs = {}
        res.pop("enc", None), res.pop("packed", None)
        res = None
        B.free()

        def note(name, fn):
            try:
                configs[name] = fn()
                # HBM-side bytes of one step of this config, where an earlier counter run left them (tools/traffic_pmc.sh)
                tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % name)
                tj = {}
                if os.path.exists(tf) and "error" not in configs[name]:
                    tj = json.load(open(tf))
                # (the same workload: the decode configs cut the data to whole blocks, 0.002 % short of 16 GiB)
                same = tj.get("input_bytes") and abs(tj["input_bytes"] - configs[name].get("input_bytes", 0)) <= 1e-3 * tj["input_bytes"]
                if os.path.exists(tf) and "error" not in configs[name] and same:
                    # (a counter run over another input size is another workload's traffic: left out)
                    configs[name]["traffic"] = tj.get("hbm_bytes_per_launch")
                    configs[name]["traffic_source"] = ("profiles/traffic_%s.json: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes over this config "
                                                       "in an earlier run (all kernels of a step; see its `reading`); not measured by this process" % name)
            except Exception as ex:                     # an extra config must not take the headline down
                configs[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            B.free()

        def encode_lv(lv):
            def run():
                r = B.encode(data, block, lv, False, xs, xw)
                out = summary(r, xs)
                out["verified"] = B.verify_encode(r, data, block)["how"]
                out["workload"] = "BGZF encode, 0xff00-byte blocks, %s, same %.2f GiB FASTQ-like data" % (
                    level_name(lv), total / 2 ** 30)
                return out
            return run
        note("encode_l2", encode_lv(2))
        note("encode_l6", encode_lv(6))

        def decode_own_l1():
            # a decode of THIS product's output: the level-1 stream of the headline data
            e = B.encode(data, block, 1, False, 1, 0)
            enc = e["enc"]
            packed = e["packed"][: e["comp_total"] + 16].clone()
            in_off = (enc.dst_off - e["dist"]["base"] + e["hdr"]).clone()
            in_len = (enc.out_len - e["hdr"]).to(torch.int32)
            want_crc = enc.crc.clone()
            del enc, e
            B.free()
            r = B.decode(data, packed, in_off, in_len, want_crc, block, xs, xw)
            out = summary(r, xs, mode="decode")
            out["workload"] = ("BGZF decode (inflate), 0xff00-byte blocks, %.2f GiB out, stream from this encoder at level 1; "
                               "output and per-block CRC-32 checked" % (total / 2 ** 30))
            return out
        note("decode_own_l1", decode_own_l1)

        def decode_ref(which, who):
            def run():
                so = os.path.join(ROOT, "oracle", "_ref", "libref.so")
                if not os.path.exists(so):
                    return {"error": "oracle/_ref/libref.so not built"}
                tb = (args.tile_mib << 20) // block * block
                nt = total // tb
                tnp = tile_np[:tb]
                d2 = torch.from_numpy(tnp).cuda().repeat(nt)
                packed, in_off, in_len, want_crc, _ = B.reference_stream(tnp, nt, block, which)
                r = B.decode(d2, packed, in_off, in_len, want_crc, block, xs, xw)
                out = summary(r, xs, mode="decode")
                out["workload"] = ("BGZF decode (inflate), 0xff00-byte blocks, %.2f GiB out, stream from the reference's "
                                   "%s (one %d MiB tile compressed on the host, its stream replicated); output and per-block "
                                   "CRC-32 checked" % (d2.numel() / 2 ** 30, who, args.tile_mib))
                return out
            return run
        # BASELINE config 3 names both reference encoders
        note("decode_libdeflate6", decode_ref("libdeflate6", "libdeflate 1.23 level 6"))
        note("decode_zlib6", decode_ref("zlib6", "zlib 1.3.1 level 6"))
        del data
        B.free()

        def migz(lv):
            def run():
                mb = 1 << 20
                _, tdata, _ = B.make_data("text", mb, whole_blocks=False)
                r = B.encode(tdata, mb, lv, True, xs, xw)
                out = summary(r, xs)
                out["verified"] = B.verify_encode(r, tdata, mb)["how"]
                out["workload"] = "MiGz encode, 1 MiB blocks, %s, %.2f GiB enwik-like text" % (
                    level_name(lv), tdata.numel() / 2 ** 30)
                return out
            return run
        note("migz_l6_text", migz(6))
        # levels 5 and 3: the same workgroup parse with two ways / one way, greedy (the speed end of the trade)
        note("migz_l5_text", migz(5))
        note("migz_l3_text", migz(3))
        line["configs"] = configs

    if rank == 0:
        if not args.no_cpu:
            line["cpu_baseline"] = cpu           # N = 1 only (None on a multi-rank run)
        print(json.dumps(line), flush=True)
    if B.use_dist:
        B.dist.destroy_process_group()


if __name__ == "__main__":
    main()
