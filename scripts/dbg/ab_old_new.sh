#!/bin/bash
# same-box A/B of the working tree against the worktree under ab_old/ (git worktree add ab_old <commit>; built): alternating rounds
# scripts/dbg/ab_old_new.sh [rounds] [models...]
set -u
R=$GRAFT_REPO_ROOT
rounds=${1:-3}; shift || true
[ $# -eq 0 ] && set -- knrm drmm drmmtks
for model in "$@"; do
  for r in $(seq $rounds); do
    for side in old new; do
      d=$R; [ $side = old ] && d=$R/ab_old
      v=$(cd $d && timeout 600 python bench.py --model $model --steps 20 --warmup 3 --no-also --no-cpu-baseline --no-pmc-traffic --no-roofline-leg --no-pass-times 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f M  %.4f ms' % (r['value']/1e6, r['ms_per_step']))")
      echo "$model $side $v"
    done
  done
done 2>&1 | tee $R/gpurun_out/ab_old_new.txt
