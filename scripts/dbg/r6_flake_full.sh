#!/bin/bash
# round 6: the whole GPU suite in a loop, the failure section of every run that has one (a test failed once in a full run and never on its own)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/flake_full.txt
for i in $(seq 1 ${1:-6}); do
  timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > /tmp/o.txt
  tail -1 /tmp/o.txt >> gpurun_out/flake_full.txt
  if grep -q " failed" /tmp/o.txt; then awk '/= FAILURES =/,/short test summary/' /tmp/o.txt | tail -120 >> gpurun_out/flake_full.txt; grep "^FAILED" /tmp/o.txt >> gpurun_out/flake_full.txt; fi
done
cut -c1-250 gpurun_out/flake_full.txt
