#!/bin/bash
# One GPU-box session: parity tests, smoke, bench variants, rocprof kernel stats.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_knrm.json; cat gpurun_out/bench_knrm.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --uniform-ids 2>/dev/null | tail -1 > gpurun_out/bench_knrm_uniform.json; cat gpurun_out/bench_knrm_uniform.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --launch-docs 1000 2>/dev/null | tail -1 > gpurun_out/bench_knrm_b1000.json; cat gpurun_out/bench_knrm_b1000.json
timeout 300 python bench.py --steps 20 --warmup 3 --model drmm 2>/dev/null | tail -1 > gpurun_out/bench_drmm.json; cat gpurun_out/bench_drmm.json
cd /tmp
P=/tmp/prof; rm -rf $P; mkdir -p $P
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/knrm -o knrm -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/drmm -o drmm -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --model drmm > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/knrm_fetch -o knrm -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/knrm_write -o knrm -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/knrm_tcc -o knrm -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/knrm_uni_fetch -o knrm -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --uniform-ids > /dev/null 2>&1
cd $R
du -sh $P
for f in $(find $P -name "*.csv" -size -2000k); do d=gpurun_out/prof/$(basename $(dirname $f)); mkdir -p $d; cp $f $d/; done
find gpurun_out/prof -name "*.csv" | head -40
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
