# on the GPU box (its copy of the tree is scratch): rebuild with -DHD_CLOCK_STAMPS and print the clock the three codec
# kernels really run at (MI355X_MICROARCH.md "DVFS give-back" item 6), each after ~2 s of back-to-back launches
set -e
cd ${GRAFT_REPO_ROOT:?run this through gpurun: it rebuilds the library with a diagnostic flag}
OUT=${1:-gpurun_out/clock}
mkdir -p $OUT
touch 7bgzf_amd/csrc/hd_api.hip
make -s -C 7bgzf_amd/csrc EXTRA=-DHD_CLOCK_STAMPS > $OUT/build.log 2>&1
: > $OUT/clock.jsonl
python3 tools/clock_stamps_child.py encode_l1 --steps 40 --warmup 2 >> $OUT/clock.jsonl
python3 tools/clock_stamps_child.py decode_libdeflate6 --mode decode --stream libdeflate6 --steps 20 --warmup 2 >> $OUT/clock.jsonl
python3 tools/clock_stamps_child.py migz_l6_text --data text --block-kib 1024 --level 6 --steps 20 --warmup 2 >> $OUT/clock.jsonl
cat $OUT/clock.jsonl
# back to the product build
touch 7bgzf_amd/csrc/hd_api.hip
make -s -C 7bgzf_amd/csrc > /dev/null 2>&1
