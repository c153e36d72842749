# round 4: the full GPU suite, the default bench line, then the round's profile set (tools/prof_round.sh), the counters of the
# workgroup parse and the traffic counters of the split-path configs; progress lines go to stdout so that a long call does not look hung
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r04_round; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout 300 -p no:cacheprovider > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
python3 bench.py > $O/bench_default.log 2>&1 || { tail -20 $O/bench_default.log; exit 1; }
grep '^{' $O/bench_default.log > $O/bench_line_default.json
python3 - $O/bench_line_default.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
print('default', j['value'], j['roofline']['frac'], j.get('verified',{}).get('how','')[:30])
for k,v in j['configs'].items(): print(' ', k, v.get('value'), v.get('ratio'), v.get('kernel_ms_avg'), v.get('error'))
print(' cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'], j['cpu_baseline'].get('per_call_adapter'))
PY
bash tools/prof_round.sh > $O/prof_round.log 2>&1 || tail -5 $O/prof_round.log
echo prof_round done
bash tools/pmc_wg.sh $O/pmc_migz6 --data text --block-kib 1024 > $O/pmc_migz6.txt 2>&1 || true
bash tools/pmc_wg.sh $O/pmc_bgzf6 > $O/pmc_bgzf6.txt 2>&1 || true
tail -2 $O/pmc_migz6.txt
bash tools/traffic_pmc.sh migz_l6_text --level 6 --data text --block-kib 1024 > $O/traffic_migz6.log 2>&1 || tail -3 $O/traffic_migz6.log
tail -1 $O/traffic_migz6.log
bash tools/traffic_pmc.sh encode_l2 --level 2 > $O/traffic_l2.log 2>&1 || tail -3 $O/traffic_l2.log
tail -1 $O/traffic_l2.log
bash tools/traffic_pmc.sh encode_l6 --level 6 > $O/traffic_l6.log 2>&1 || tail -3 $O/traffic_l6.log
tail -1 $O/traffic_l6.log
bash tools/pmc_inflate.sh $O/pmc_inflate > $O/pmc_inflate.txt 2>&1 || true
tail -1 $O/pmc_inflate.txt
