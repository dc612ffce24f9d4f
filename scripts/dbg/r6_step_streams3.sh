#!/bin/bash
# round 6: step streams on the int32 candidate-store route and with two batches in rotation
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for spec in "--resident --step-streams 1" "--resident --step-streams 2" "--batches 2 --step-streams 1" "--batches 2 --step-streams 2" "--step-streams 2" "--resident --batches 4 --step-streams 2"; do
  v=$(timeout 600 python bench.py --model knrm --steps 20 --warmup 4 --repeats 3 $spec --no-also --no-cpu-baseline --no-pmc-traffic --no-roofline-leg --no-pass-times 2>gpurun_out/ss_err.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f M  %.4f ms  (min %.4f max %.4f) serial %s' % (r['value']/1e6, r['ms_per_step'], r['repeats']['ms_per_step_min'], r['repeats']['ms_per_step_max'], r.get('step_streams',{}).get('serial_steps',{}).get('ms_per_step')))" 2>&1 | tail -1)
  echo "knrm $spec: $v"
done 2>&1 | tee gpurun_out/step_streams_resident.txt
