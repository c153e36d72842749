# usage: bash tools/exp_inflate_ss.sh [fuzz]  -- the self-synchronising batch decoder (-DHD_INFLATE_SS): inflate tests, decode
# rates, optionally the wide inflate fuzz (experiment: rebuilds the library with the flag, the guard rebuilds the default)
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
. tools/exp_guard.sh
exp_guard
mkdir -p gpurun_out/ss
touch 7bgzf_amd/csrc/hd_api.hip
make -s -C 7bgzf_amd/csrc EXTRA=-DHD_INFLATE_SS > gpurun_out/ss/make.log 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "inflate or decode or unpipe or round or dictzip or razf or ciso or png" > gpurun_out/ss/tests.log 2>&1 || true
tail -5 gpurun_out/ss/tests.log
for st in libdeflate6 zlib6 own; do
  python bench.py --mode decode --stream $st --steps 3 --warmup 1 --no-cpu 2>gpurun_out/ss/bench_$st.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('decode $st', j['value'], 'GB/s kernel ms', j['roofline']['kernel_ms_avg'])" || tail -3 gpurun_out/ss/bench_$st.err
done
if [ "$1" = fuzz ]; then
  timeout -k 10 900 python tools/big_fuzz_inflate.py > gpurun_out/ss/fuzz_inflate.log 2>&1 || true
  tail -3 gpurun_out/ss/fuzz_inflate.log
fi
touch 7bgzf_amd/csrc/hd_api.hip
