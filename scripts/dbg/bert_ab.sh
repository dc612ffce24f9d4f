#!/bin/bash
run() { env "$@" python bench.py --model bert --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$*', 'docs/s', round(r['value'],1), 'ffn1 us', round(ro['kernel_ms']*1e3,1), 'step nominal frac', round(ro['whole_step_frac_nominal'],4))"; }
for rep in 1 2; do
run X=1
run CAPAMD_GEMM_NGROUP=12
run CAPAMD_GEMM_NGROUP=6
run CAPAMD_GEMM_NGROUP=3
run CAPAMD_GEMM_NGROUP=2
run CAPAMD_GEMM_NGROUP=9
run CAPAMD_GEMM_NGROUP=1
done
