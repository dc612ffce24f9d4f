#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for round in 1 2; do
for v in base ckabl1 ckabl2 ckabl4 ckabl6; do
  libenv="X=1"; [ $v != base ] && libenv="CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_$v.so"
  env $libenv CAPAMD_BENCH_NO_CHECK=1 CAPAMD_BENCH_NOCHECK=1 python bench.py --model convknrm --no-cpu-baseline --no-roofline-leg --no-pmc-traffic --no-pass-times 2>gpurun_out/err_$v.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,2), 'M pairs/s', round(d['ms_per_step'],4), 'ms')"
done
done
