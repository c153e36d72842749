# experiment: how much of the emit-only kernel is code construction (token loop disabled; output is wrong on purpose)
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/exp_guard.sh
exp_guard 7bgzf_amd/csrc/hd_deflate_dynamic.hpp
export TMPDIR=/tmp
python3 - <<'PY'
p='7bgzf_amd/csrc/hd_deflate_dynamic.hpp'; s=open(p).read()
s=s.replace("			for (uint32_t base = 0; base < ntok_slab; base += 64) {\n				const uint32_t k = base + lane;\n				const bool valid","			for (uint32_t base = 0; base < (EMIT ? 0u : ntok_slab); base += 64) {\n				const uint32_t k = base + lane;\n				const bool valid")
open(p,'w').write(s)
PY
make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_exp -- python3 bench.py --steps 2 --warmup 1 --level 2 --no-cpu > gpurun_out/kt_exp.log 2>&1 || true
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/kt_exp/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'hd::k_deflate' in r['Name']:
            print(r['Name'][:62], r['Calls'], round(float(r['TotalDurationNs'])/1e6/3,1),'ms per step')
PY
