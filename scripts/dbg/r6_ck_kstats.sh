#!/bin/bash
# ConvKNRM list route vs per-pair: kernel averages (scripts/dbg/r6_ck_kstats.sh)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for x in "" "--per-pair"; do
rm -rf /tmp/p
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $R/bench.py --model convknrm $x --steps 8 --warmup 2 --no-cpu-baseline --no-pass-times > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("== convknrm $x")
for r in rows[:8]:
    print("  %-70s calls %5s avg us %9.1f total ms %8.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done 2>&1 | tee $R/gpurun_out/r6_ck_kstats.txt
