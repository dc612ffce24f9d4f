#!/bin/bash
# One GPU-box session: parity tests, smoke, bench lines, rocprof kernel stats + PMC traffic.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_knrm.json; cut -c1-400 gpurun_out/bench_knrm.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --uniform-ids 2>/dev/null | tail -1 > gpurun_out/bench_knrm_uniform.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --launch-docs 1000 2>/dev/null | tail -1 > gpurun_out/bench_knrm_b1000.json
timeout 300 python bench.py --steps 20 --warmup 3 --model drmm 2>/dev/null | tail -1 > gpurun_out/bench_drmm.json
timeout 600 python bench.py --steps 3 --warmup 1 --model bert 2>/dev/null | tail -1 > gpurun_out/bench_bert.json
timeout 600 python bench.py --steps 3 --warmup 1 --model bert --no-cpu-baseline --bert-skip-padding 2>/dev/null | tail -1 > gpurun_out/bench_bert_skip_padding.json
for mdl in drmmtks pacrr convknrm; do timeout 300 python bench.py --steps 20 --warmup 3 --model $mdl 2>/dev/null | tail -1 > gpurun_out/bench_$mdl.json; done
timeout 600 python scripts/sibling_bench.py 2>/dev/null | grep model > gpurun_out/bench_siblings.jsonl; cat gpurun_out/bench_siblings.jsonl
for f in knrm_uniform knrm_b1000 drmm bert drmmtks pacrr convknrm; do python -c "import json;r=json.load(open('gpurun_out/bench_$f.json'));print('$f', round(r['value'],1), r['roofline']['frac'])"; done
cd /tmp; P=/tmp/prof; rm -rf $P; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/knrm -o knrm -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/drmm -o drmm -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --model drmm > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/bert -o bert -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --model bert > /dev/null 2>&1
for mdl in drmmtks pacrr convknrm; do timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$mdl -o $mdl -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --model $mdl > /dev/null 2>&1; done
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/knrm_fetch -o knrm -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/knrm_write -o knrm -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/knrm_tcc -o knrm -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/knrm_uni_fetch -o knrm -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --uniform-ids > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/drmm_fetch -o drmm -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --model drmm > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/bert_mfma -o bert -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --model bert --docs 256 > /dev/null 2>&1
cd $R
timeout 200 python scripts/gemm_bench.py 2>/dev/null | tail -5 > gpurun_out/gemm_bench.txt; cat gpurun_out/gemm_bench.txt
bash scripts/pmc_gemm_vs_vendor.sh > /dev/null 2>&1; cat gpurun_out/gemm_vs_vendor.txt | cut -c1-220
for f in $(find $P -name "*.csv" -size -3000k); do d=gpurun_out/prof/$(basename $(dirname $f)); mkdir -p $d; cp $f $d/; done
ls gpurun_out/prof/*/ | head -40
