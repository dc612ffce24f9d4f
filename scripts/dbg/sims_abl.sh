#!/bin/bash
# sims pass, ablation builds side by side (rocprofv3 kernel averages of a short DRMM bench run): scripts/dbg/sims_abl.sh lib1 lib2 ...   ("base" = the product library)
set -u
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for name in "$@"; do
  libenv="X=1"; [ "$name" != base ] && libenv="CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_$name.so"
  rm -rf /tmp/p
  env $libenv CAPAMD_BENCH_NO_CHECK=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline-leg --no-pass-times --model ${MODEL:-drmm} > /dev/null 2>/tmp/err.txt
  python - <<PY
import csv,glob
try:
    f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    print("== %-10s" % "$name", "  ".join("%s %.1f" % (r["Name"].split("lists_")[-1][:16], float(r["AverageNs"])/1e3) for r in rows if int(r["Calls"]) >= 10 and "lists_" in r["Name"]))
except Exception as e:
    print("== $name FAILED", e, open("/tmp/err.txt").read()[-800:])
PY
done 2>&1 | tee $R/gpurun_out/sims_abl.txt
