#!/bin/bash
# Round 5, session 1: BERT parity tests on the new epilogues, then an interleaved A/B of the encoder (round-4 library vs this tree; micro-batch plans).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_bert.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8 ) > gpurun_out/pytest_bert.log 2>&1; cat gpurun_out/pytest_bert.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --model bert --steps 4 --warmup 2 --no-cpu-baseline --no-bert-other-dtype 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$tag', 'docs/s', round(r['value'],1), 'ms', round(r['ms_per_step'],2), 'ffn1 us', round(ro['kernel_ms']*1e3,1), 'frac exec', round(ro['whole_step_frac'],4), 'nominal', round(ro['whole_step_frac_nominal'],4))"; }
for rep in 1 2 3; do
run r4 CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_r4.so
run new_equal CAPAMD_BERT_MB_PLAN=equal
run new_full X=1
done 2>&1 | tee gpurun_out/r5_bert_ab.txt
cd /tmp; P=/tmp/prof; rm -rf $P; mkdir -p $P
KS="rocprofv3 --kernel-trace --stats --output-format csv"
timeout 300 $KS -d $P/bert -o bert -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bert-other-dtype --bert-streams 1 --model bert > /dev/null 2>&1
CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_r4.so timeout 300 $KS -d $P/bert_r4 -o bert -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bert-other-dtype --bert-streams 1 --model bert > /dev/null 2>&1
cd $R
for d in bert bert_r4; do f=$(find $P/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${d}_kernel_stats.csv && head -8 $f | cut -c1-200; done
