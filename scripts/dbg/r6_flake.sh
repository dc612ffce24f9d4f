#!/bin/bash
# round 6: hunting a test that failed once in a full GPU run (a training-graph test): the graph / fused-step tests in a loop
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/flake.txt
for i in $(seq 1 ${1:-10}); do
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "graph or fused or train" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > /tmp/o.txt
  tail -1 /tmp/o.txt >> gpurun_out/flake.txt
  if grep -q "failed" /tmp/o.txt; then grep -B 60 "short test summary" /tmp/o.txt | tail -90 >> gpurun_out/flake.txt; fi
done
cat gpurun_out/flake.txt | cut -c1-250
