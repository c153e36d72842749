set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/prof1/bench_kt.log 2>&1
echo kt_done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/prof1/pmc1 -- python3 bench.py --steps 1 --warmup 0 --gib 2 --no-cpu > gpurun_out/prof1/bench_pmc1.log 2>&1
echo pmc1_done
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/prof1/pmc2 -- python3 bench.py --steps 1 --warmup 0 --gib 2 --no-cpu > gpurun_out/prof1/bench_pmc2.log 2>&1
echo pmc2_done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof1/pmc3 -- python3 bench.py --steps 1 --warmup 0 --gib 2 --no-cpu > gpurun_out/prof1/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof1/pmc4 -- python3 bench.py --steps 1 --warmup 0 --gib 2 --no-cpu > gpurun_out/prof1/bench_pmc4.log 2>&1
echo pmc34_done
find gpurun_out/prof1 -name "*.csv" | head -30
du -sh gpurun_out/prof1
