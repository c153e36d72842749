# encode rate and ratio per level on the two BENCH data kinds (16 GiB, 0xff00-byte BGZF blocks) and the MiGz 1 MiB text case
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
run() { python3 bench.py --no-cpu --no-extra --steps ${STEPS:-3} --warmup 1 "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', j['value'], 'GB/s  ratio', j['config']['ratio'], ' kernel ms', j['roofline']['kernel_ms_avg'])"; }
for l in ${LEVELS:-2 6 9}; do
  run --level $l
  run --level $l --data text
done
run --level 6 --data text --block-kib 1024
run --level 9 --data text --block-kib 1024
