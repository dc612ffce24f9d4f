#!/bin/bash
# lists route, library variants side by side: scripts/dbg/lists_variants.sh name[@libname] ...   (two alternating rounds per model, then kernel averages)
set -u
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for model in knrm drmm; do
  for round in 1 2; do
    for spec in "$@"; do
      name=${spec%%[:@]*}; lib=""; envs=""
      case "$spec" in *@*) lib=${spec##*@};; esac
      case "$spec" in *:*) envs=${spec#*:}; envs=${envs%%@*}; envs=${envs//,/ };; esac
      libenv="X=1"; [ -n "$lib" ] && libenv="CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
      extra=""; [ $name = pairs ] && extra="--per-pair"
      env $envs $libenv timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline-leg --model $model $extra 2>gpurun_out/lists_err.txt | tail -1 > gpurun_out/ab_$name.json
      python - <<PY
import json
try:
    r = json.load(open("gpurun_out/ab_$name.json"))
    print("%-5s %-8s %7.2f M pairs/s  %.4f ms" % ("$model", "$name", r["value"] / 1e6, r["ms_per_step"]))
except Exception as e:
    print("$model $name FAILED", e); print(open("gpurun_out/lists_err.txt").read()[-1500:])
PY
    done
  done
done 2>&1 | tee gpurun_out/lists_variants.txt
cd /tmp
for spec in "$@"; do
  name=${spec%%[:@]*}; lib=""; envs=""
  case "$spec" in *@*) lib=${spec##*@};; esac
  case "$spec" in *:*) envs=${spec#*:}; envs=${envs%%@*}; envs=${envs//,/ };; esac
  [ $name = pairs ] && continue
  libenv="X=1"; [ -n "$lib" ] && libenv="CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
  for model in knrm drmm; do
    rm -rf /tmp/p
    env $envs $libenv timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline-leg --model $model > /dev/null 2>&1
    python - <<PY
import csv,glob
f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("== $name $model", "  ".join("%s %.1f" % (r["Name"].split("lists_")[-1][:14] if "lists_" in r["Name"] else r["Name"][:14], float(r["AverageNs"])/1e3) for r in rows[:14] if int(r["Calls"]) >= 10 and "lists_" in r["Name"]))
PY
  done
done 2>&1 | tee -a $R/gpurun_out/lists_variants.txt
