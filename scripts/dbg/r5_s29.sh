#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-also --no-roofline-leg --no-pmc-traffic --no-pass-times "$@" 2>gpurun_out/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', round(d['value']/1e6,2), 'M pairs/s', round(d['ms_per_step'],4), 'ms')" || tail -3 gpurun_out/err.txt; }
for r in 1 2; do
run --model drmm
run --model drmm --launch-docs 125000 --launch-streams 2
run --model drmm --launch-docs 84000 --launch-streams 3
done
