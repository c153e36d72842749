"""what hip_inflate (the zlibutil_code_dec boundary, lib/zlibutil.h:46) costs per 0xff00-byte block, from 1 / 16 / 64 calling
threads -- the way `7bgzf -d -@N` calls it (applet/7bgzf.c:330-345: a thread per block).  A DEFLATE stream is one wavefront's
serial work however empty the chip is; concurrent callers share a launch (hd_api.hip, inflate_one).
usage: python tools/inflate_call_latency.py [out.jsonl]   (on the GPU box; drives 7bgzf_amd/inflate_call_bench)"""
import importlib, json, os, subprocess, sys, tempfile, zlib
sys.path.insert(0, '.')
pkg = importlib.import_module("7bgzf_amd")
synth = importlib.import_module("7bgzf_amd.synth")
blk = bytes(synth.fastq_like(0xff00, seed=5))
exe = os.path.join(os.path.dirname(pkg.LIB_PATH), "inflate_call_bench")
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for name, z in (("zlib6", zlib.compress(blk, 6)[2:-4]), ("own_level1", pkg.hip_deflate(blk, 1)[1])):
    r, back = pkg.hip_inflate(z, 0xff00)
    assert r == 0 and back == blk
    with tempfile.NamedTemporaryFile(suffix=".deflate") as f:
        f.write(z)
        f.flush()
        for threads in (1, 4, 16, 64):
            for env_extra in ({},) + (({"HIPDEFLATE_DEVICES": "0,0"},) if threads == 64 else ()):
                p = subprocess.run([exe, f.name, str(0xff00), str(threads), "2"], capture_output=True, text=True,
                                   env=dict(os.environ, **env_extra))
                assert p.returncode == 0, p.stderr[-2000:]
                rec = dict(json.loads(p.stdout.strip().splitlines()[-1]), stream=name, **env_extra)
                print(json.dumps(rec))
                if out:
                    out.write(json.dumps(rec) + "\n")
