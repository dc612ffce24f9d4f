#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15 ) > gpurun_out/pytest_gpu.log 2>&1; tail -18 gpurun_out/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --model bert --steps 4 --warmup 2 --no-cpu-baseline --no-bert-other-dtype 2>gpurun_out/err_$tag.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$tag', 'docs/s', round(r['value'],1), 'ms', round(r['ms_per_step'],2), 'ffn1 us', round(ro['kernel_ms']*1e3,1), 'frac exec', round(ro['whole_step_frac'],4), 'nominal', round(ro['whole_step_frac_nominal'],4))" || tail -5 gpurun_out/err_$tag.txt; }
for rep in 1 2; do
run default X=1
run r4_mix CAPAMD_GEMM_PICK=qkv=128,ffn1=128,oproj=256x32,ffn2=256x32
done 2>&1 | tee gpurun_out/bert_ab.txt
PYTHONPATH=$R timeout 300 python scripts/dbg/ring16_timeline.py 2>&1 | grep "resid" | tee gpurun_out/ring16_timeline_resid.txt
