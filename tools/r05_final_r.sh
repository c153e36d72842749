# round 5: last check of the tree as it is committed -- smoke() and the default bench line as the driver runs it
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_final_r; mkdir -p $O
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 400 python3 bench.py > $O/bench.log 2>&1 || { tail -10 $O/bench.log; exit 1; }
grep '^{' $O/bench.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['metric'], j['value'], j['unit'], 'frac', j['roofline']['frac'], 'cpu', j['cpu_baseline']['value'], {k: v.get('value') for k, v in j['configs'].items()})"
