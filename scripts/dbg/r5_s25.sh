#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "lists or list or multiquery or resident or predict" 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --no-also --no-roofline-leg --no-pmc-traffic --no-pass-times "$@" 2>gpurun_out/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', round(d['value']/1e6,2), 'M pairs/s', round(d['ms_per_step'],4), 'ms')" || tail -3 gpurun_out/err.txt; }
for r in 1 2; do
run --model knrm
run --model drmm
run --model drmmtks
run --model pacrr
run --model drmmtks --queries 250
run --model pacrr --queries 250
done
