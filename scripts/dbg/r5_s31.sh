#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "knrm or drmm or lists or multiquery or resident or similarity" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2
for round in 1 2 3; do
for v in base dpr0; do
  libenv="X=1"; [ $v != base ] && libenv="CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_$v.so"
  for model in drmm; do
  env $libenv python bench.py --model $model --no-cpu-baseline --no-also --no-roofline-leg --no-pmc-traffic --no-pass-times 2>gpurun_out/err_$v.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v $model', round(d['value']/1e6,2), 'M pairs/s', round(d['ms_per_step'],4), 'ms')"
  done
done
done
