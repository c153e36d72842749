# round 5, measurement set A: the default bench line, then the round's profile set (tools/prof_round.sh)
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r05_round; mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench_default.log 2>&1 || { tail -20 $O/bench_default.log; exit 1; }
grep '^{' $O/bench_default.log > $O/bench_line_default.json
python3 - $O/bench_line_default.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
print('default', j['value'], j['roofline']['frac'], j.get('verified',{}).get('how','')[:30])
for k,v in j['configs'].items(): print(' ', k, v.get('value'), v.get('ratio'), v.get('kernel_ms_avg'), v.get('error'))
print(' cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'], j['cpu_baseline'].get('per_call_adapter'))
PY
timeout -k 10 700 bash tools/prof_round.sh > $O/prof_round.log 2>&1 || tail -5 $O/prof_round.log
echo prof_round done
timeout -k 10 300 python3 bench.py --first-record 1 --no-extra --no-cpu > $O/bench_first_record_1.log 2>&1 || tail -5 $O/bench_first_record_1.log
grep '^{' $O/bench_first_record_1.log > $O/bench_first_record_1.json
python3 -c "
import json; j=json.load(open('$O/bench_first_record_1.json')); print('first-record 1:', j['value'], j['config']['ratio'], j['roofline']['kernel_ms_avg'])"
