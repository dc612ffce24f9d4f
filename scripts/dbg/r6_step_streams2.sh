#!/bin/bash
# round 6: which call size / stream count a long run should use: bench.py --queries N --step-streams S (KNRM, DRMM), alternating rounds on one box
set -u
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
for r in 1 2; do
  for spec in "knrm 64 1" "knrm 64 2" "knrm 128 1" "knrm 128 2" "knrm 256 1" "knrm 256 2" "knrm 32 2" "knrm 32 3" "drmm 64 1" "drmm 64 2" "drmm 125 1" "drmm 125 2" "drmm 250 1"; do
    set -- $spec
    v=$(timeout 600 python bench.py --model $1 --queries $2 --steps 12 --warmup 4 --repeats 3 --step-streams $3 --no-also --no-cpu-baseline --no-pmc-traffic --no-roofline-leg --no-pass-times 2>gpurun_out/ss_err.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f M  %.4f ms  (min %.4f max %.4f)' % (r['value']/1e6, r['ms_per_step'], r['repeats']['ms_per_step_min'], r['repeats']['ms_per_step_max']))" 2>&1 | tail -1)
    echo "$1 lists/step=$2 streams=$3 $v"
  done
done 2>&1 | tee gpurun_out/step_streams_sizes.txt
