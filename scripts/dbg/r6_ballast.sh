#!/bin/bash
# round 6: the sims pass capped at 4 / 3 / 2 workgroups per CU (dynamic-LDS ballast builds) under two step streams: does the other stream's
# mark / pooling pass co-reside with it?   (scripts/build_variant_obj.sh lists ballastNN -DCAPAMD_LISTS_SIMS_BALLAST=NN first)
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for r in 1 2; do
  for lib in "" ballast23 ballast31 ballast50; do
    for S in 1 2; do
      libenv="X=1"; [ -n "$lib" ] && libenv="CAPAMD_LIB_PATH=$GRAFT_REPO_ROOT/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
      v=$(env $libenv timeout 600 python bench.py --model knrm --steps 20 --warmup 4 --repeats 3 --step-streams $S --no-also --no-cpu-baseline --no-pmc-traffic --no-roofline-leg --no-pass-times 2>gpurun_out/ss_err.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f M  %.4f ms  (min %.4f max %.4f)' % (r['value']/1e6, r['ms_per_step'], r['repeats']['ms_per_step_min'], r['repeats']['ms_per_step_max']))" 2>&1 | tail -1)
      echo "knrm lib=${lib:-default} streams=$S $v"
    done
  done
done 2>&1 | tee gpurun_out/ballast_ab.txt
