# on the GPU box (its copy of the tree is scratch): rebuild with -DHD_INFLATE_STATS and print where k_inflate's tokens and cycles go
set -e
cd ${GRAFT_REPO_ROOT:?run this through gpurun: it rebuilds the library with an experiment flag}
touch 7bgzf_amd/csrc/hd_api.hip
make -s -C 7bgzf_amd/csrc EXTRA=-DHD_INFLATE_STATS > /dev/null 2>&1
bash tools/exp_inflate_stats.sh
