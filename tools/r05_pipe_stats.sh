cd ${GRAFT_REPO_ROOT:?}
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make EXTRA=-DHD_INFLATE_STATS ../libhipdeflate.so > /dev/null 2>&1)
timeout -k 10 200 python3 tools/exp_inflate_pipe_stats.py 2>&1 | tail -3 | tee gpurun_out/r05_pipe_stats.txt
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make ../libhipdeflate.so > /dev/null 2>&1)
