#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bert.py -m gpu -q -x -k "gemm" 2>&1 | tail -4
ONLY16=1 PYTHONPATH=$R timeout 300 python scripts/dbg/ring16_timeline.py 2>&1 | grep "ring" | grep -v qkv | tee gpurun_out/ring16_timeline_128.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --model bert --steps 4 --warmup 2 --no-cpu-baseline --no-bert-other-dtype 2>gpurun_out/err_$tag.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$tag', 'docs/s', round(r['value'],1), 'ms', round(r['ms_per_step'],2), 'ffn1 us', round(ro['kernel_ms']*1e3,1), 'frac exec', round(ro['whole_step_frac'],4), 'nominal', round(ro['whole_step_frac_nominal'],4))" || tail -5 gpurun_out/err_$tag.txt; }
for rep in 1 2; do
run default X=1
run ffn1_128x16 CAPAMD_GEMM_PICK=ffn1=128x16
run ffn1_prod_128x16 CAPAMD_GEMM_PICK=ffn1=128x16,oproj=128x16,ffn2=128x16
run oproj_128x16 CAPAMD_GEMM_PICK=oproj=128x16
done 2>&1 | tee gpurun_out/bert_ab128.txt
