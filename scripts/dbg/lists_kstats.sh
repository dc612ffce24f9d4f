#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for model in knrm drmm; do
rm -rf /tmp/p
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline-leg --model $model > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("== $model")
for r in rows[:9]:
    if "lists_" in r["Name"] or "Memset" in r["Name"] or "fill" in r["Name"].lower(): print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
