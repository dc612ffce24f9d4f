# rocprofv3 kernel stats of the BERT bench leg (top kernels), builder-side helper
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o bert -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --model bert "$@" > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:8]: print(r["Name"][:72], r["Calls"], round(float(r["AverageNs"])/1e3,1), round(100*float(r["TotalDurationNs"])/tot,1))
PY
