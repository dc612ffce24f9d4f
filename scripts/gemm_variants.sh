#!/bin/bash
# A/B of profiling builds of the library (capreolus_amd/csrc/ablate/*.so) on the K-step probe and the GEMM bench
for v in "" "$@"; do
  if [ -z "$v" ]; then echo "== default build"; unset CAPAMD_LIB_PATH; else echo "== $v"; export CAPAMD_LIB_PATH=$PWD/capreolus_amd/csrc/ablate/libcapreolus_amd_$v.so; fi
  timeout 200 python scripts/gemm_kstep_probe.py 2>&1 | grep -E "M=65536 N=2304|M=2048  N=2304"
  timeout 200 python scripts/gemm_bench.py 2>&1 | grep -E "qkv-like|ffn1" | cut -c1-70
done
