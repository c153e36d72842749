# experiment: variants of the workgroup parse built on the box, MiGz level 6 and BGZF level 3 only (the verification of a timing-only variant may fail: --no-verify is not offered, so its line is simply absent)
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=$PWD/gpurun_out/exp_wgq; mkdir -p $O
for V in $HD_WG_VARIANTS; do
  name=${V%%:*}; flags=""; [ "$V" != "$name" ] && flags=${V#*:}
  (cd 7bgzf_amd/csrc && rm -f hd_api.o && make EXTRA="$flags" ../libhipdeflate.so > $O/build_$name.log 2>&1) || { tail -5 $O/build_$name.log; exit 1; }
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -- python3 bench.py --steps 2 --warmup 1 --level 6 --data text --block-kib 1024 --gib 4 --no-cpu --no-extra > $O/$name.log 2>&1 || true
  echo "$name: $(find $O/kt_$name -name '*kernel_stats.csv' | head -1 | xargs grep k_parse_wg | cut -d, -f1-4)"
done
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make ../libhipdeflate.so > /dev/null 2>&1)
