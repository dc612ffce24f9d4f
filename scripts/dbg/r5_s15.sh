#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bert.py -m gpu -q -x 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --model bert --steps 4 --warmup 2 --no-cpu-baseline --no-bert-other-dtype 2>gpurun_out/err_$tag.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$tag', 'docs/s', round(r['value'],1), 'ms', round(r['ms_per_step'],2), 'ffn1 us', round(ro['kernel_ms']*1e3,1), 'frac exec', round(ro['whole_step_frac'],4), 'nominal', round(ro['whole_step_frac_nominal'],4))" || tail -5 gpurun_out/err_$tag.txt; }
for rep in 1 2 3; do
run default X=1
run nopf CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_nopf.so
run all16 CAPAMD_GEMM_PICK=ffn1=256x16
run all16_nopf CAPAMD_GEMM_PICK=ffn1=256x16 CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_nopf.so
done 2>&1 | tee gpurun_out/bert_pf_ab.txt
