set -e
cd ${GRAFT_REPO_ROOT:-.}
python3 -c "
import importlib,sys
sys.path.insert(0,'.')
s=importlib.import_module('7bgzf_amd.synth')
s.fastq_like(64<<20, seed=1234).tofile('/tmp/hook_fq.bin')
"
for M in hip6 hip3; do for T in 4 8 16 32 64; do
  for F in latency throughput; do
    HIPDEFLATE_HOOK_FORM=$F BGZF_METHOD=$M ./7bgzf_amd/hook_bench /tmp/hook_fq.bin $T 2 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$M $F T=$T', j['GBps_in'], j['ratio'], j['us_per_call'])"
  done
done; done
