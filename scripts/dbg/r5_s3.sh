#!/bin/bash
cd scripts/ubench
for sh in "65536 2304 768" "65536 768 768" "65536 3072 768" "65536 768 3072"; do
  for rep in 1 2; do
  echo "32x32x16: $(./gemm4w2 $sh)"
  echo "16x16x32: $(./gemm4w5 $sh)"
  done
done 2>&1 | tee ../../gpurun_out/gemm4w5_vs_4w2.txt
