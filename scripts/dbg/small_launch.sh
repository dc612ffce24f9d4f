#!/bin/bash
for st in 1 2 4 6 8; do
  for ld in 1000 500; do
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-roofline-leg --launch-docs $ld --launch-streams $st 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('graph streams $st launch $ld', round(r['value']/1e6,2), 'M pairs/s', round(r['ms_per_step'],3))"
  done
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-roofline-leg --launch-docs 1000 --model drmm 2>&1 | tail -1 | cut -c1-200
python -m pytest tests/test_gpu_parity.py -q -k "predict or full_size or ndcg" 2>&1 | tail -3
