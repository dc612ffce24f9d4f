#!/bin/bash
# whole-list route against the per-pair kernels: parity tests, then the KNRM and DRMM headline legs both ways (two alternating rounds)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "lists" 2>&1 | tail -15
for model in knrm drmm; do
  for round in 1 2; do
    for mode in lists pairs; do
      extra=""; [ $mode = pairs ] && extra="--per-pair"
      timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline-leg --model $model $extra 2>gpurun_out/lists_err.txt | tail -1 > gpurun_out/ab_$mode.json
      python - <<PY
import json
try:
    r = json.load(open("gpurun_out/ab_$mode.json"))
    h = r["roofline"]["headline_leg"]
    print("%-5s %-6s %7.2f M pairs/s  %.4f ms   requested %.0f GB/s  %s" % ("$model", "$mode", r["value"] / 1e6, r["ms_per_step"], h["requested_GBps"], h.get("mean_distinct_terms_per_list", "")))
except Exception as e:
    print("$model $mode FAILED", e); print(open("gpurun_out/lists_err.txt").read()[-1500:])
PY
    done
  done
done 2>&1 | tee gpurun_out/lists_ab.txt
