#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ONLY16=1 PYTHONPATH=$R timeout 200 python scripts/dbg/ring16_timeline.py 2>&1 | grep "resid" | tee gpurun_out/ring16_timeline_early.txt
timeout 900 python -m pytest tests/test_gpu_bert.py -m gpu -q -x 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --model bert --steps 4 --warmup 2 --no-cpu-baseline --no-bert-other-dtype 2>gpurun_out/err_$tag.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$tag', 'docs/s', round(r['value'],1), 'ms', round(r['ms_per_step'],2), 'ffn1 us', round(ro['kernel_ms']*1e3,1), 'frac exec', round(ro['whole_step_frac'],4), 'nominal', round(ro['whole_step_frac_nominal'],4))" || tail -5 gpurun_out/err_$tag.txt; }
for rep in 1 2; do
run default X=1
run r4_mix CAPAMD_GEMM_PICK=qkv=128,ffn1=128,oproj=256x32,ffn2=256x32
done 2>&1 | tee gpurun_out/bert_ab.txt
