# round 5, measurement set B: traffic counters of the split-path configs and of decode, the counters of the workgroup parse and of inflate
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r05_round; mkdir -p $O
for spec in "migz_l6_text --level 6 --data text --block-kib 1024" "encode_l2 --level 2" "encode_l6 --level 6" "decode_libdeflate6 --mode decode --stream libdeflate6"; do
  set -- $spec; name=$1; shift
  timeout -k 10 400 bash tools/traffic_pmc.sh $name "$@" > $O/traffic_$name.log 2>&1 || tail -3 $O/traffic_$name.log
  tail -1 $O/traffic_$name.log | cut -c1-200
done
timeout -k 10 300 bash tools/pmc_wg.sh $O/pmc_migz6 --data text --block-kib 1024 > $O/pmc_migz6.txt 2>&1 || true
tail -2 $O/pmc_migz6.txt | cut -c1-300
timeout -k 10 300 bash tools/pmc_inflate.sh $O/pmc_inflate > $O/pmc_inflate.txt 2>&1 || true
tail -1 $O/pmc_inflate.txt | cut -c1-300
