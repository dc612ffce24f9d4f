#!/bin/bash
# Round 5, session 2: which tile shape each encoder GEMM wants with the lighter epilogues; ring timeline; predict end to end.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --model bert --steps 4 --warmup 2 --no-cpu-baseline --no-bert-other-dtype 2>gpurun_out/err_$tag.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$tag', 'docs/s', round(r['value'],1), 'ms', round(r['ms_per_step'],2), 'ffn1 us', round(ro['kernel_ms']*1e3,1), 'frac exec', round(ro['whole_step_frac'],4), 'nominal', round(ro['whole_step_frac_nominal'],4))" || tail -5 gpurun_out/err_$tag.txt; }
for rep in 1 2; do
run default X=1
run all256 CAPAMD_RING_BM=256
run all128 CAPAMD_RING_BM=128
run pingpong CAPAMD_GEMM_RING=0
run streams1 CAPAMD_BERT_STREAMS=1
run streams3 CAPAMD_BERT_STREAMS=3
run r4 CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_r4.so
done 2>&1 | tee gpurun_out/r5_s2_ab.txt
PYTHONPATH=$R timeout 300 python scripts/dbg/ring_timeline.py 2>&1 | tail -14 | tee gpurun_out/ring_timeline.txt
PYTHONPATH=$R timeout 300 python $R/scripts/predict_e2e_bench.py 2>/dev/null | tail -1 > gpurun_out/predict_e2e.json; cat gpurun_out/predict_e2e.json
