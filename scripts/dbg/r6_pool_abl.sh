#!/bin/bash
# round 6: what the KNRM pooling pass spends its time on - kernel averages of ablation builds (ids without a load, lookups folded into 16 KB, both)
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for lib in default "$@"; do
  libenv="X=1"; [ $lib != default ] && libenv="CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
  rm -rf /tmp/p; env $libenv CAPAMD_BENCH_NO_CHECK=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline-leg --no-pass-times --step-streams 1 > /dev/null 2>/tmp/err.txt
  python - <<PY
import csv,glob
f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)
if not f: print("lib=$lib FAILED", open("/tmp/err.txt").read()[-600:])
else:
    print("lib=%-10s" % "$lib", "  ".join("%s %.1f" % (r["Name"].split("lists_")[1].split("(")[0][:16], float(r["AverageNs"])/1e3) for r in csv.DictReader(open(f[0])) if "lists_" in r["Name"]))
PY
done 2>&1 | tee $R/gpurun_out/pool_abl.txt
