#!/bin/bash
# round 6: KNRM list-route variants, one bench round + kernel averages each   (scripts/dbg/r6_knrm_variants.sh name[@lib] ...)
set -u
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for spec in "$@"; do
  name=${spec%%[:@]*}; lib=""; envs=""
  case "$spec" in *@*) lib=${spec##*@};; esac
  case "$spec" in *:*) envs=${spec#*:}; envs=${envs%%@*}; envs=${envs//,/ };; esac
  libenv="X=1"; [ -n "$lib" ] && libenv="CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
  rm -rf /tmp/p
  env $envs $libenv CAPAMD_BENCH_NO_CHECK=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline-leg --no-pass-times --model ${MODEL:-knrm} > /tmp/out.txt 2>/tmp/err.txt
  python - <<PY
import csv,glob,json
try:
    r = json.loads(open("/tmp/out.txt").read().strip().splitlines()[-1]); v = "%.1f M pairs/s %.3f ms" % (r["value"]/1e6, r["ms_per_step"])
except Exception as e:
    v = "bench failed: " + open("/tmp/err.txt").read()[-600:]
f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0]))) if f else []
print("== %-8s %s |" % ("$name", v), "  ".join("%s %.1f" % (r["Name"].split("lists_")[-1][:12], float(r["AverageNs"])/1e3) for r in rows[:14] if int(r["Calls"]) >= 10 and "lists_" in r["Name"]))
PY
done 2>&1 | tee -a $R/gpurun_out/r6_knrm_variants.txt
