# usage: bash tools/save_profiles.sh rNN   -- copy the summaries of gpurun_out/prof_round (tools/prof_round.sh) into profiles/
set -e
cd "$(dirname "$0")/.."
R=$1; P=gpurun_out/prof_round
latest() { ls -t $1 | head -1; }       # (gpurun merges into the local directory: older calls leave their files there)
cp $(latest "$P/kt/*/*_kernel_stats.csv") profiles/${R}_encode_l1_kernel_stats.csv
cp $(latest "$P/kt_l2/*/*_kernel_stats.csv") profiles/${R}_encode_l2_kernel_stats.csv
cp $(latest "$P/kt_l6/*/*_kernel_stats.csv") profiles/${R}_encode_l6_kernel_stats.csv
cp $(latest "$P/kt_migz6/*/*_kernel_stats.csv") profiles/${R}_encode_migz_l6_kernel_stats.csv
cp $(latest "$P/kt_dec/*/*_kernel_stats.csv") profiles/${R}_decode_libdeflate6_kernel_stats.csv
cp $P/bench_line.json profiles/${R}_bench_line_under_rocprof.json
cp $P/bench_line_migz6.json profiles/${R}_bench_line_migz_l6_under_rocprof.json
cp $P/bench_line_dec.json profiles/${R}_bench_line_decode_under_rocprof.json
cp $P/traffic_encode_l1.json profiles/traffic_encode_l1.json
[ -f $P/pmc_l1/summary.json ] && cp $P/pmc_l1/summary.json profiles/${R}_encode_l1_pmc_summary.json
ls -la profiles | grep ${R}_ | wc -l
