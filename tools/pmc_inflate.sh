# usage: bash tools/pmc_inflate.sh <outdir> [stream]   -- two SQ counter passes over a 2 GiB decode of the reference's stream:
# instructions per 64 output bytes and where the waves wait
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export TMPDIR=/tmp
OUT=$1; S=${2:-libdeflate6}
mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc1 -- python3 bench.py --mode decode --stream $S --steps 1 --warmup 0 --gib 2 --tile-mib 16 --no-cpu > $OUT/b1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc2 -- python3 bench.py --mode decode --stream $S --steps 1 --warmup 0 --gib 2 --tile-mib 16 --no-cpu > $OUT/b2.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,collections,sys,json
out=sys.argv[1]
acc=collections.defaultdict(float)
for f in glob.glob(out+'/pmc*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_inflate' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value'])
units=(2<<30)/64
res={k:round(v/units,2) for k,v in acc.items()}
res['SQ_WAVE_CYCLES']=round(res.get('SQ_WAVE_CYCLES',0)/2,2)     # collected in both passes
json.dump({"per_64_output_bytes":res,"note":"quad-cycle units for the *_CYCLES / WAIT / ACTIVE counters"},open(out+'/inflate_pmc.json','w'),indent=1)
print(res)
PY
