# Which pipe binds the KNRM headline kernel: LDS-array cycles, VALU / LDS / VMEM issue activity, wait buckets (counters only; one pass per group)
cd /tmp; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also --no-roofline-leg --no-pmc-traffic"
for grp in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pk; timeout 300 rocprofv3 --output-format csv --pmc $grp -d /tmp/pk -o c -- $B "$@" > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pk/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])) if f else []:
    if "knrm_forward" in r["Kernel_Name"] or "drmm_forward" in r["Kernel_Name"] or "stream_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v)) for k, v in acc.items()}, "launches", max([len(v) for v in acc.values()] or [0]))
PY
done
