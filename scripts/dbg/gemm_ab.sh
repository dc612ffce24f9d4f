#!/bin/bash
# two passes over all variants (clock / box drift shows as the difference between the passes)
for pass in 1 2; do
for v in "" "$@"; do
  if [ -z "$v" ]; then unset CAPAMD_LIB_PATH; v=default; else export CAPAMD_LIB_PATH=$PWD/capreolus_amd/csrc/ablate/libcapreolus_amd_$v.so; fi
  timeout 200 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids | awk -v v=$v '{printf "%s %s=%s ", (NR==1? v ":" : ""), $1, $(NF-3)} END {print ""}'
done
done
