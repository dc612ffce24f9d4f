#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 0 1 2 4 8 3 15; do
  if [ $n = 0 ]; then unset CAPAMD_PROF_LIB_PATH; else export CAPAMD_PROF_LIB_PATH=$R/capreolus_amd/csrc/ablate/libprof_r16abl$n.so; fi
  echo "abl=$n $(ONLY=oproj ONLY16=1 PYTHONPATH=$R timeout 200 python scripts/dbg/ring16_timeline.py 2>&1 | grep resid)"
done | tee gpurun_out/resid16_ablation.txt
