import importlib, sys, zlib
sys.path.insert(0, '.')
synth = importlib.import_module("7bgzf_amd.synth")
blk = bytes(synth.fastq_like(0xff00, seed=5))
open("gpurun_out/z6.deflate", "wb").write(zlib.compress(blk, 6)[2:-4])
