# usage: bash tools/exp_hook_queues.sh  -- experiment: does the hook's device phase depend on how HIP maps the batches' streams
# to hardware queues (GPU_MAX_HW_QUEUES) or on the spin/sleep policy of the members?
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
export HIPDEFLATE_HOOK_STATS=1 BGZF_METHOD=${1:-hip2}
run() { echo "== $*"; env "$@" ./7bgzf_amd/hook_bench /tmp/hook_fq.bin 16 2 2>&1 | sed 's/.*us per batch: //; s/"block.*"GBps_in"/ GBps/' | cut -c1-160; }
run A=0
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
run HIPDEFLATE_SPIN_US=0
run HIPDEFLATE_SPIN_US=400
run HIP_FORCE_DEV_KERNARG=1
run AMD_DIRECT_DISPATCH=0
