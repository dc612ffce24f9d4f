#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for ng in 1 2 3 4 6 12; do
  echo "== CAPAMD_GEMM_NGROUP=$ng"
  GEMM_ONLY=FFN1 CAPAMD_GEMM_NGROUP=$ng PYTHONPATH=$R timeout 300 python scripts/gemm_bench_ring.py 64000 bf16 2>/dev/null | grep FFN1
done 2>&1 | tee gpurun_out/ffn1_ngroup_sweep.txt
for st in 64 256; do
  echo "== CAPAMD_RING_STAGGER=$st"
  GEMM_ONLY=FFN1 CAPAMD_RING_STAGGER=$st PYTHONPATH=$R timeout 300 python scripts/gemm_bench_ring.py 64000 bf16 2>/dev/null | grep FFN1
done 2>&1 | tee -a gpurun_out/ffn1_ngroup_sweep.txt
