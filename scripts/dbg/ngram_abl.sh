#!/bin/bash
# ngram_conv.hip, ablation builds side by side (rocprofv3 kernel averages of scripts/ngram_conv_bench.py): scripts/dbg/ngram_abl.sh [--full] lib1 lib2 ...  ("base" = the product library)
set -u
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
extra=""; [ "$1" = "--full" ] && { extra="--full"; shift; }
for name in "$@"; do
  libenv="X=1"; [ "$name" != base ] && libenv="CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_$name.so"
  rm -rf /tmp/p
  env $libenv timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $R/scripts/ngram_conv_bench.py $extra > /tmp/out.txt 2>/tmp/err.txt
  python - <<PY
import csv,glob
try:
    f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    print("== %-10s" % "$name", "  ".join("%s %.1f" % (r["Name"].split("ngram_")[-1][:18], float(r["AverageNs"])/1e3) for r in rows if "ngram_" in r["Name"]))
except Exception as e:
    print("== $name FAILED", e, open("/tmp/err.txt").read()[-800:])
PY
done 2>&1 | tee -a $R/gpurun_out/ngram_abl.txt
