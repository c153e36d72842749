# usage: bash tools/prof_pmc.sh <outdir> [bench args...]   -- SQ counter passes on a 2 GiB run
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc1 -- python3 bench.py --steps 1 --warmup 0 --gib 2 --tile-mib 16 --no-cpu "$@" > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 bench.py --steps 1 --warmup 0 --gib 2 --tile-mib 16 --no-cpu "$@" > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED --output-format csv -d $OUT/pmc3 -- python3 bench.py --steps 1 --warmup 0 --gib 2 --tile-mib 16 --no-cpu "$@" > $OUT/bench_pmc3.log 2>&1 || true
python3 - $OUT <<'PY'
import csv,glob,collections,sys,json
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out+'/pmc*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        if 'hd::' in r['Kernel_Name']:
            acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
json.dump(acc,open(out+'/summary.json','w'),indent=1)
for k,v in acc.items(): print(k,dict(v))
PY
