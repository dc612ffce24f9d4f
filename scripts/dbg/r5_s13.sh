#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { tag=$1; shift; extra=""; case "$tag" in mb128*) extra="--bert-microbatch 128";; mb192*) extra="--bert-microbatch 192";; esac; env "$@" timeout 300 python bench.py --model bert --steps 4 --warmup 2 --no-cpu-baseline --no-bert-other-dtype $extra 2>gpurun_out/err_$tag.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$tag', 'docs/s', round(r['value'],1), 'ms', round(r['ms_per_step'],2), 'ffn1 us', round(ro['kernel_ms']*1e3,1), 'frac exec', round(ro['whole_step_frac'],4), 'nominal', round(ro['whole_step_frac_nominal'],4))" || tail -5 gpurun_out/err_$tag.txt; }
for rep in 1 2; do
run default X=1
run mb128 X=1
run mb128_share2 CAPAMD_GEMM_CU_SHARE=2
run mb128_share2_s4 CAPAMD_GEMM_CU_SHARE=2 CAPAMD_BERT_STREAMS=4
run mb192 X=1
done 2>&1 | tee gpurun_out/bert_mb_ab.txt
