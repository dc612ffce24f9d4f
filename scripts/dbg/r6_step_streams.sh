#!/bin/bash
# round 6: consecutive steps round-robin over S HIP streams (bench.py --step-streams S) against one stream, alternating rounds on one box
# scripts/dbg/r6_step_streams.sh [rounds] [models...]
set -u
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
rounds=${1:-2}; shift || true
[ $# -eq 0 ] && set -- knrm drmm drmmtks
for model in "$@"; do
  for r in $(seq $rounds); do
    for S in 1 2 3; do
      v=$(timeout 600 python bench.py --model $model --steps 20 --warmup 4 --step-streams $S --no-also --no-cpu-baseline --no-pmc-traffic --no-roofline-leg --no-pass-times 2>gpurun_out/ss_err.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f M  %.4f ms  (min %.4f max %.4f)' % (r['value']/1e6, r['ms_per_step'], r['repeats']['ms_per_step_min'], r['repeats']['ms_per_step_max']))" 2>&1 | tail -1)
      echo "$model streams=$S $v"
    done
  done
done 2>&1 | tee gpurun_out/step_streams_ab.txt
tail -5 gpurun_out/ss_err.txt
