#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "pacrr" 2>&1 | tail -4
python bench.py --model pacrr --no-cpu-baseline --no-roofline-leg --no-pmc-traffic 2>&1 | tail -1 > gpurun_out/pacrr_q4.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/pacrr_q4.json').read())
print(d['value'], d['ms_per_step'])
for p in d['roofline'].get('passes', []): print(p.get('kernel','')[:50], p.get('ms'))
PY
python bench.py --model pacrr --per-pair --no-cpu-baseline --no-roofline-leg --no-pmc-traffic 2>&1 | tail -1 | cut -c1-200
