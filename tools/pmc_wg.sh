# usage: bash tools/pmc_wg.sh <outdir> [bench args...]  -- two SQ counter passes over a 2 GiB level-6 run: instructions per 64-byte
# step of the workgroup parse (k_parse_wg) and of the emit kernel, and where the wavefronts wait
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --gib 2 --tile-mib 16 --no-cpu --no-extra --level 6 $*"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc1 -- python3 bench.py $ARGS > $OUT/b1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc2 -- python3 bench.py $ARGS > $OUT/b2.log 2>&1
python3 - $OUT "$ARGS" <<'PY'
import csv,glob,collections,sys,json
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out+'/pmc*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'k_parse_wg' in k or 'k_deflate_dynamic' in k:
            acc['k_parse_wg' if 'k_parse_wg' in k else 'k_deflate_dynamic<EMIT>'][r['Counter_Name']]+=float(r['Counter_Value'])
units=(2<<30)/64
res={}
for k,v in acc.items():
    res[k]={c:round(x/units,2) for c,x in v.items()}
    res[k]['SQ_WAVE_CYCLES']=round(res[k].get('SQ_WAVE_CYCLES',0)/2,2)     # collected in both passes
json.dump({"bench_args":sys.argv[2],"per_64_input_bytes":res,"note":"quad-cycle units for the *_CYCLES / WAIT / ACTIVE counters"},open(out+'/wg_pmc.json','w'),indent=1)
for k,v in res.items(): print(k,v)
PY
