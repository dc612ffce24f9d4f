#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for cfg in "CAPAMD_RING_BM=128 CAPAMD_RING_STAGGER=0" "CAPAMD_GEMM_RING=0"; do
  rm -rf /tmp/pp
  env $cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o b -- python $R/bench.py --model bert --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    n=r['Name']
    n=n.replace('_ZN6capamd','').replace('EvNS_8GemmArgsE','').replace('(anonymous namespace)::','')
    print(f"{n[:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
done
cd $R; python -m pytest tests/test_gpu_bert.py -q -x -k "ring_gemm_small" 2>&1 | grep -E "Mismatch|Greatest|passed|failed|FAILED" | head
