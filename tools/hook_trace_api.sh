# usage: bash tools/hook_trace_api.sh <method> <callers>  -- HIP runtime calls + kernels of the hook under T callers: what the
# leader's hipdeflate_lat_run spends beside the kernels.  Output under gpurun_out/hook_trace/.
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/hook_trace
mkdir -p $OUT
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
M=${1:-hip2}; T=${2:-8}
export BGZF_METHOD=$M HIPDEFLATE_HOOK_STATS=1
ROOT=$PWD
(cd /tmp && rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d /tmp/ha_${M}_$T -o ha -- $ROOT/7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 > $OUT/${M}_T${T}_api.log 2>&1) || tail -5 $OUT/${M}_T${T}_api.log
for f in $(find /tmp/ha_${M}_$T -name "*stats.csv"); do cp $f $OUT/${M}_T${T}_$(basename $f); done
grep -h "hipdeflate hook\|GBps" $OUT/${M}_T${T}_api.log | cut -c1-400
cut -c1-170 $OUT/${M}_T${T}_ha_hip_api_stats.csv 2>/dev/null || ls $OUT
