# experiment: variants of the workgroup parse built on the box (EXTRA=-D...), each timed on the three level-6 workloads
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=$PWD/gpurun_out/exp_wg; mkdir -p $O
for V in ${HD_WG_VARIANTS:-base static:-DHD_WG_EXP_STATIC_DEAL}; do
  name=${V%%:*}; flags=""; [ "$V" != "$name" ] && flags=${V#*:}
  (cd 7bgzf_amd/csrc && rm -f hd_api.o && make EXTRA="$flags" ../libhipdeflate.so > $O/build_$name.log 2>&1) || { tail -5 $O/build_$name.log; exit 1; }
  for cfg in "migz6 --data text --block-kib 1024" "bgzf6 " "bgzf6_text --data text"; do
    set -- $cfg; c=$1; shift
    python3 bench.py --steps 3 --warmup 1 --level 6 --no-cpu --no-extra "$@" > $O/${name}_$c.log 2>&1 || tail -5 $O/${name}_$c.log
    grep '^{' $O/${name}_$c.log | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('$name $c', j['value'], j['ms_per_step'], j['config']['ratio'])
"
  done
done
(cd 7bgzf_amd/csrc && rm -f hd_api.o && make ../libhipdeflate.so > /dev/null 2>&1)
