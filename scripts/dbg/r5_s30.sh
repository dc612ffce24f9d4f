#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base mabl1 mabl2 mabl3; do
  libenv="X=1"; [ $v != base ] && libenv="CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_$v.so"
  rm -rf /tmp/p
  env $libenv CAPAMD_BENCH_NO_CHECK=1 CAPAMD_BENCH_NOCHECK=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $R/bench.py --model knrm --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-roofline-leg --no-pmc-traffic --no-pass-times > /dev/null 2>&1
  f=$(find /tmp/p -name "*kernel_stats.csv" | head -1)
  python - "$f" $v <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
import re
print(sys.argv[2], "  ".join(f"{re.search(r'lists_[a-z_]+', r['Name']).group(0)} {float(r['AverageNs'])/1e3:.1f}" for r in rows if 'lists_' in r['Name']))
PY
done
