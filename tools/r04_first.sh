set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
python3 bench.py --steps 3 --warmup 1 --level 6 --data text --block-kib 1024 --no-cpu --no-extra > $O/migz6.log 2>&1
python3 bench.py --steps 3 --warmup 1 --level 6 --no-cpu --no-extra > $O/bgzf6.log 2>&1
python3 bench.py --steps 3 --warmup 1 --level 6 --data text --no-cpu --no-extra > $O/bgzf6_text.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_migz6 -- python3 bench.py --steps 2 --warmup 1 --level 6 --data text --block-kib 1024 --no-cpu --no-extra > $O/kt_migz6.log 2>&1
python3 bench.py > $O/default.log 2>&1
tail -n 3 $O/*.log
