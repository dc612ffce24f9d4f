#!/bin/bash
# quick record of the round's state: the default bench line, the BERT line, rocprofv3 kernel stats of the BERT leg
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py"
mkdir -p gpurun_out
( time timeout 900 $B --steps 20 --warmup 5 2>gpurun_out/err_default.txt | tail -1 > gpurun_out/bench_default.json ) 2>&1 | grep real; cp bench_full.json gpurun_out/bench_full_default.json 2>/dev/null; cut -c1-1500 gpurun_out/bench_default.json
timeout 600 $B --steps 5 --warmup 2 --model bert 2>/dev/null | tail -1 > gpurun_out/bench_bert.json; cp bench_full.json gpurun_out/bench_full_bert.json 2>/dev/null; cut -c1-900 gpurun_out/bench_bert.json
cd /tmp; P=/tmp/prof; rm -rf $P; mkdir -p $P
KS="rocprofv3 --kernel-trace --stats --output-format csv"
timeout 300 $KS -d $P/bert -o bert -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-bert-other-dtype --bert-streams 1 --model bert > /dev/null 2>&1
cd $R
f=$(find $P/bert -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/bert_bench_kernel_stats.csv && head -9 $f | cut -c1-160
