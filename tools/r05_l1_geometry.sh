# round 5 (secondary): level 1 -- static Huffman, one wavefront per block -- with a wider window: what it buys on text (the
# gap to slz) and what it costs in throughput (the ring and the table are the wavefront's LDS: occupancy).  Each variant = library AND twin
# rebuilt with the geometry, bench.py's own verification (kernel == twin, round trip) on, level 1 on the FASTQ-like set and on text
set -o pipefail
cd ${GRAFT_REPO_ROOT:?}
O=gpurun_out/r05_l1_geometry; mkdir -p $O; : > $O/ab.txt
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], 'GB/s, ratio', j['config'].get('ratio'), 'verified', j['verified'].get('how', j['verified']) if isinstance(j.get('verified'), dict) else j.get('verified'))"; }
for g in ${GEOMETRIES:-12,11 13,12 14,12 15,13 12,11}; do
  D="-DHD_L1_WIN_BITS=${g%,*} -DHD_L1_HASH_BITS=${g#*,}"
  touch 7bgzf_amd/csrc/hd_api.hip oracle/hd_deflate_twin.c
  make -s -C 7bgzf_amd/csrc EXTRA="$D" > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
  make -s -C oracle CC="gcc $D" > $O/build_oracle.log 2>&1 || { tail -5 $O/build_oracle.log; exit 1; }
  echo "== window 2^${g%,*}, table geometry ${g#*,}: $(grep -A14 "k_deflate_staticILi${g%,*}ELi${g#*,}ELb0ELi4ELi0ELi0ELi0ELb0" 7bgzf_amd/csrc/hd_api.resources.log | grep -E 'LDS Size' | head -1 | sed 's/.*remark: *//')" | tee -a $O/ab.txt
  timeout -k 10 150 python3 bench.py --level 1 --no-cpu --steps 5 --warmup 1 --no-extra 2>$O/err.log | line l1_fastq | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
  timeout -k 10 150 python3 bench.py --level 1 --data text --no-cpu --steps 5 --warmup 1 --no-extra 2>$O/err.log | line l1_text | tee -a $O/ab.txt || { tail -3 $O/err.log; exit 1; }
done
touch 7bgzf_amd/csrc/hd_api.hip oracle/hd_deflate_twin.c
make -s -C 7bgzf_amd/csrc > /dev/null 2>&1; make -s -C oracle > /dev/null 2>&1
