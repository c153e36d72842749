# usage: bash tools/hook_trace.sh <method> <callers>  -- kernel trace of the hook under T callers (hook_bench): are the kernels of a
# batch as long as in hipdeflate_lat_run alone (tools/lat_trace.sh)?  Output under gpurun_out/hook_trace/.
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
./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 > $OUT/${M}_T${T}_plain.txt 2>&1
rm -rf /tmp/ht_${M}_$T
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ht_${M}_$T -o ht -- $ROOT/7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 > $OUT/${M}_T${T}_rocprof.log 2>&1) || tail -5 $OUT/${M}_T${T}_rocprof.log
cp "$(find /tmp/ht_${M}_$T -name "*kernel_stats.csv" | head -1)" $OUT/${M}_T${T}_kernel_stats.csv
cut -c1-400 $OUT/${M}_T${T}_plain.txt; grep -h "hook\|GBps" $OUT/${M}_T${T}_rocprof.log | cut -c1-400; cut -c1-160 $OUT/${M}_T${T}_kernel_stats.csv
