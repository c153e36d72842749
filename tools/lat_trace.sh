# usage: bash tools/lat_trace.sh [level ...]  -- kernel trace of hipdeflate_lat_run alone (hook_bench <file> 0): which kernel
# holds a latency-mode batch.  Output: gpurun_out/lat_trace/L<level>_kernel_stats.csv + the run's own timings.
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/lat_trace
mkdir -p $OUT
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
for lv in ${@:-1 2 6}; do
  export HOOK_LEVEL=$lv
  unset HOOK_N
  ./7bgzf_amd/hook_bench /tmp/hook_fq.bin 0 > $OUT/L${lv}_run.txt
  export HOOK_N=${HOOK_TRACE_N:-16}
  ROOT=$PWD
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt_$lv -o lt -- $ROOT/7bgzf_amd/hook_bench /tmp/hook_fq.bin 0 > $OUT/L${lv}_rocprof.log 2>&1) || tail -5 $OUT/L${lv}_rocprof.log
  cp "$(find /tmp/lt_$lv -name "*kernel_stats.csv" | head -1)" $OUT/L${lv}_kernel_stats.csv
  echo "== level $lv"; cat $OUT/L${lv}_run.txt | cut -c1-150; cut -c1-200 $OUT/L${lv}_kernel_stats.csv
done
