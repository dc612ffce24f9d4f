#!/bin/bash
# Round 5, session 4: the 16x16x32 ring kernel - parity tests, then the encoder A/B (which GEMMs want it).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_bert.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25 ) > gpurun_out/pytest_bert.log 2>&1; tail -30 gpurun_out/pytest_bert.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --model bert --steps 4 --warmup 2 --no-cpu-baseline --no-bert-other-dtype 2>gpurun_out/err_$tag.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$tag', 'docs/s', round(r['value'],1), 'ms', round(r['ms_per_step'],2), 'ffn1 us', round(ro['kernel_ms']*1e3,1), 'frac exec', round(ro['whole_step_frac'],4), 'nominal', round(ro['whole_step_frac_nominal'],4), 'err', r.get('parity',{}).get('max_score_error_of_scale_vs_fp32_port'))" || tail -5 gpurun_out/err_$tag.txt; }
for rep in 1 2; do
run default X=1
run all256_ring16 CAPAMD_RING_BM=256
run ring16_off CAPAMD_RING16=0
done 2>&1 | tee gpurun_out/r5_s4_ab.txt
cd /tmp; P=/tmp/prof; rm -rf $P; mkdir -p $P
KS="rocprofv3 --kernel-trace --stats --output-format csv"
timeout 300 $KS -d $P/bert -o bert -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bert-other-dtype --bert-streams 1 --model bert > /dev/null 2>&1
CAPAMD_RING_BM=256 timeout 300 $KS -d $P/bert_all256 -o bert -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bert-other-dtype --bert-streams 1 --model bert > /dev/null 2>&1
cd $R
for d in bert bert_all256; do f=$(find $P/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${d}_kernel_stats.csv && head -8 $f | cut -c1-150; done
