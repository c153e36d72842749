# usage: bash tools/exp_inflate_ss_stats.sh [G ...]  -- counters of the self-synchronising batch decoder (experiment build, on the box)
set -e
cd ${GRAFT_REPO_ROOT:?run this through gpurun: it rebuilds the library with experiment flags}
for G in ${@:-0}; do
  touch 7bgzf_amd/csrc/hd_api.hip
  FIX=""; [ "$G" != 0 ] && FIX="-DSS_FIXG=$G"
  make -s -C 7bgzf_amd/csrc EXTRA="-DHD_INFLATE_SS -DHD_INFLATE_STATS $FIX" > /dev/null 2>&1
  echo "== G $G"
  python3 tools/exp_inflate_stats_child.py libdeflate6 2>/dev/null | tail -2
done
echo "SS build: windows=batches, window_tokens=tokens listed, scalar_tokens=lanes in pass 2 (first round), eob=... that met their first pass, slow_litlen=sum of G, slow_dist=batches ending in a stop, window_empty=sum of V, '-'=sum of NL"
