#!/bin/bash
# BERT bench leg with / without the per-device GEMM kernel choice; the choices are printed (CAPAMD_GEMM_AUTOTUNE_VERBOSE)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
for at in 1 0 1 0; do
  CAPAMD_GEMM_AUTOTUNE=$at CAPAMD_GEMM_AUTOTUNE_VERBOSE=1 timeout 600 python bench.py --steps 5 --warmup 2 --model bert --no-cpu-baseline --no-bert-other-dtype 2> gpurun_out/bert_tune_$at.err | tail -1 > gpurun_out/bench_bert_tune$at.json
  grep capamd gpurun_out/bert_tune_$at.err | sort | uniq | head -8
  python -c "
import json; r=json.load(open('gpurun_out/bench_bert_tune$at.json')); print('autotune $at', round(r['value'],1), 'docs/s', round(r['ms_per_step'],2), 'ms executed', round(r['roofline']['whole_step_frac'],4), 'nominal', round(r['roofline']['whole_step_frac_nominal'],4), 'ffn1 us', round(r['roofline']['kernel_ms']*1e3,1))"
done
