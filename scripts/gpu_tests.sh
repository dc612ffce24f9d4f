#!/bin/bash
# The whole GPU suite + smoke, as the driver runs them at round end; output under gpurun_out/pytest_gpu.log
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q "$@" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25 ) > gpurun_out/pytest_gpu.log 2>&1; cat gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
