# experiment: cycles of the emit-only kernel by phase on the workgroup levels' records (a stats build made on the box)
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make EXTRA=-DHD_EMIT_STATS ../libhipdeflate.so > /dev/null 2>&1)
python3 tools/exp_emit_stats.py --level 6 --data text --block-kib 1024 2>&1 | tail -3
python3 tools/exp_emit_stats.py --level 6 --data text 2>&1 | tail -3
python3 tools/exp_emit_stats.py --level 6 2>&1 | tail -3
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make ../libhipdeflate.so > /dev/null 2>&1)
