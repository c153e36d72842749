# decode rate on the three streams the inflate kernel is tuned on: the reference's libdeflate-6 and zlib-6 streams of the
# FASTQ-like set (BASELINE config 3) and our own level-1 stream (short matches, more tokens per byte)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
for s in libdeflate6 zlib6; do
  python3 bench.py --mode decode --stream $s --no-cpu --steps ${STEPS:-5} --warmup 2 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$s', j['value'], 'GB/s out, kernel ms', j['roofline']['kernel_ms_avg'])"
done
python3 bench.py --mode decode --stream own --level 1 --no-cpu --steps ${STEPS:-5} --warmup 2 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('own-level-1', j['value'], 'GB/s out, kernel ms', j['roofline']['kernel_ms_avg'])"
