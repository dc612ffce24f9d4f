#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bert.py -m gpu -q -x 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --model bert --steps 4 --warmup 2 --no-cpu-baseline --no-bert-other-dtype 2>gpurun_out/err_$tag.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$tag', 'docs/s', round(r['value'],1), 'ms', round(r['ms_per_step'],2), 'ffn1 us', round(ro['kernel_ms']*1e3,1), 'frac exec', round(ro['whole_step_frac'],4), 'nominal', round(ro['whole_step_frac_nominal'],4))" || tail -5 gpurun_out/err_$tag.txt; }
for rep in 1 2 3; do
run default X=1
run oldattn CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_oldattn.so
done 2>&1 | tee gpurun_out/bert_attn_ab.txt
cd /tmp; rm -rf /tmp/p; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bert-other-dtype --bert-streams 1 --model bert > /dev/null 2>&1; f=$(find /tmp/p -name "*kernel_stats.csv" | head -1); grep "attention_s256" $f | cut -c1-160
