#!/bin/bash
# A/B of KNRM / DRMM kernel builds on one box: scripts/dbg/knrm_stream_ab.sh [model] name[:ENV=V,...][@libname] ...
#   name alone = the in-tree library; @libname = capreolus_amd/csrc/ablate/libcapreolus_amd_<libname>.so (scripts/build_variant_obj.sh)
# Parity of the streaming kernel first (CAPAMD_KNRM_STREAM=2 forces it wherever the geometry allows), then two alternating rounds of the bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
model=knrm
case "${1:-}" in knrm|drmm) model=$1; shift;; esac
( CAPAMD_KNRM_STREAM=2 CAPAMD_DRMM_STREAM=2 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "knrm or KNRM or drmm or DRMM or random_geometries or full_size or resident or permut" 2>&1 | tail -8 ) > gpurun_out/stream_parity.log 2>&1; tail -3 gpurun_out/stream_parity.log
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-pmc-traffic --model $model"
for round in 1 2; do
  for spec in "$@"; do
    name=${spec%%[:@]*}; envs=""; lib=""
    case "$spec" in *@*) lib=${spec##*@};; esac
    case "$spec" in *:*) envs=${spec#*:}; envs=${envs%%@*}; envs=${envs//,/ };; esac
    libenv=""; [ -n "$lib" ] && libenv="CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
    env $envs $libenv timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/ab_$name.json
    python - <<PY
import json
try:
    r = json.load(open("gpurun_out/ab_$name.json"))
    print("%-14s %7.2f M pairs/s  %.4f ms   hbm-leg frac %s" % ("$name", r["value"] / 1e6, r["ms_per_step"], r["roofline"].get("frac")))
except Exception as e:
    print("$name FAILED", e)
PY
  done
done
