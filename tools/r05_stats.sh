cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
OUT=gpurun_out/r05_stats
mkdir -p $OUT
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make EXTRA=-DHD_EMIT_STATS ../libhipdeflate.so > /dev/null 2>&1)
for k in fastq text; do timeout -k 10 120 python3 tools/exp_emit_wg_stats.py 6 $k 2>&1 | tail -1 | tee -a $OUT/emit_wg_stats.txt || exit 1; done
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make ../libhipdeflate.so > /dev/null 2>&1)
