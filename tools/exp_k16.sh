# usage: bash tools/exp_k16.sh   -- A/B of the K16 / continuation-lane parse per level and data kind (experiment only:
# rebuilds the library on the box with -DHD_K16_OFF, then restores the default build)
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out
. tools/exp_guard.sh
exp_guard        # no source is patched; the trap rebuilds the default library whatever happens
run_set() {
  for data in fastq text; do
    for lv in 2 3 5 6 9; do
      python bench.py --steps 2 --warmup 1 --no-cpu --no-extra --gib 4 --tile-mib 32 --level $lv --data $data 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1 $data L$lv', j['value'], 'GB/s ratio', j['config']['ratio'])"
    done
  done
}
run_set k16
[ "$1" = only ] && exit 0
touch 7bgzf_amd/csrc/hd_api.hip
make -s -C 7bgzf_amd/csrc EXTRA=-DHD_K16_OFF > /dev/null 2>&1
run_set off
touch 7bgzf_amd/csrc/hd_api.hip      # (the trap's make must not think the -DHD_K16_OFF objects are current)
