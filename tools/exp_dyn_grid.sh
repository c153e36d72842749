# experiment: persistent-grid width of the dynamic levels (run on the GPU box; edits the working copy there only)
set -e
cd $GRAFT_REPO_ROOT
for g in "$@"; do
  sed -i "s/	const uint32_t per_cu = level >= 5 ? 4u : [0-9]*u;/	const uint32_t per_cu = level >= 5 ? 4u : ${g}u;/" 7bgzf_amd/csrc/hd_deflate_dynamic.hpp
  make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
  echo "== per_cu $g"
  python bench.py --steps 2 --warmup 1 --no-cpu --gib 8 --tile-mib 32 --level 3 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  ', j['value'], 'GB/s kernel ms', j['roofline']['kernel_ms_avg'])"
done
