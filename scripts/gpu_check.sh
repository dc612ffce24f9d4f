#!/bin/bash
# Quick GPU-box session: parity tests, smoke, the default bench line.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/pytest_gpu.log 2>&1; cat gpurun_out/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/bench_default.log 2>&1; tail -c 6000 gpurun_out/bench_default.log
