# usage: bash tools/pmc_quick.sh <outdir> [bench args...]   -- ONE SQ counter pass over a 2 GiB run: instructions per 64-byte step
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
rocprofv3 --pmc ${HD_PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA} --output-format csv -d $OUT/pmc -- python3 bench.py --steps 1 --warmup 0 --gib 2 --tile-mib 16 --no-cpu --no-extra "$@" > $OUT/bench_pmc.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,collections,sys,json
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out+'/pmc/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'hd::' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0][-48:]][r['Counter_Name']]+=float(r['Counter_Value'])
steps=(2<<30)/64
res={}
for k,v in acc.items():
    res[k]={c:round(x/steps,2) for c,x in v.items()}
    res[k]['_raw']=dict(v)
json.dump(res,open(out+'/summary.json','w'),indent=1)
for k,v in res.items():
    print(k,{c:x for c,x in v.items() if c!='_raw'})
PY
