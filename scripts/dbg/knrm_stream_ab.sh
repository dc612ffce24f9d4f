#!/bin/bash
# A/B of the streaming KNRM kernel (persistent workgroups + list wave) against the one-pair-per-workgroup kernel: parity first, then the bench legs.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( CAPAMD_KNRM_STREAM=2 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "knrm or KNRM or random_geometries or full_size or resident or permut" 2>&1 | tail -8 ) > gpurun_out/stream_parity.log 2>&1; cat gpurun_out/stream_parity.log
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-pmc-traffic"
for mode in 1 0 1 0; do
  CAPAMD_KNRM_STREAM=$mode timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/bench_knrm_stream$mode.json
  python - <<PY
import json
r = json.load(open("gpurun_out/bench_knrm_stream$mode.json"))
print("stream=$mode", round(r["value"] / 1e6, 2), "M pairs/s", r["ms_per_step"], "ms  roofline frac", r["roofline"].get("frac"), r["roofline"].get("headline_leg", {}))
PY
done
