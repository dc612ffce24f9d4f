#!/bin/bash
# Copies what scripts/gpu_round.sh left under gpurun_out/ (scratch) into profiles/<round>/ (tracked).  Usage: collect_profiles.sh r01
set -eu
D=profiles/${1:-r01}
G=gpurun_out
mkdir -p $D
for f in bench_knrm bench_knrm_uniform bench_knrm_b1000 bench_drmm bench_bert bench_bert_skip_padding bench_drmmtks bench_pacrr bench_convknrm; do cp $G/$f.json $D/; done
cp $G/bench_siblings.jsonl $D/
cp $G/prof/knrm/knrm_kernel_stats.csv $D/knrm_bench_kernel_stats.csv
cp $G/prof/drmm/drmm_kernel_stats.csv $D/drmm_bench_kernel_stats.csv
cp $G/prof/bert/bert_kernel_stats.csv $D/bert_bench_kernel_stats.csv
for m in drmmtks pacrr convknrm; do [ -f $G/prof/$m/${m}_kernel_stats.csv ] && cp $G/prof/$m/${m}_kernel_stats.csv $D/${m}_bench_kernel_stats.csv; done
cp $G/prof/knrm_fetch/knrm_counter_collection.csv $D/knrm_fetch_counters.csv
cp $G/prof/knrm_write/knrm_counter_collection.csv $D/knrm_write_counters.csv
cp $G/prof/knrm_tcc/knrm_counter_collection.csv $D/knrm_tcc_counters.csv
cp $G/prof/knrm_uni_fetch/knrm_counter_collection.csv $D/knrm_uni_fetch_counters.csv
cp $G/prof/drmm_fetch/drmm_counter_collection.csv $D/drmm_fetch_counters.csv
cp $G/prof/bert_mfma/bert_counter_collection.csv $D/bert_mfma_counters.csv
cp $G/gemm_bench.txt $D/gemm_bench_vs_hipblaslt.txt
cp $G/gemm_vs_vendor.txt $D/gemm_pmc_vs_hipblaslt.txt
ls -la $D | tail -30
