#!/bin/bash
# Same-box A/B of the whole-forward MFMA-busy fraction and wall clock: LayerNorm folded into the GEMMs (1) vs separate passes (0)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for c in 0 1 0 1; do
  rm -rf /tmp/pab; CAPAMD_BERT_FUSED_LN=$c timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pab -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --model bert --docs 256 > /dev/null 2>&1
  f=$(find /tmp/pab -name "*counter_collection.csv")
  python - "$f" $c <<'PY'
import csv, collections, sys
d = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    v = d[r['Dispatch_Id']]; v[r['Counter_Name']] = float(r['Counter_Value']); v['dur'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
vs = [v for v in d.values() if 'GRBM_GUI_ACTIVE' in v and 'SQ_VALU_MFMA_BUSY_CYCLES' in v]
g = sum(v['GRBM_GUI_ACTIVE'] for v in vs) / 8; b = sum(v['SQ_VALU_MFMA_BUSY_CYCLES'] for v in vs); dur = sum(v['dur'] for v in vs)
print("fused_ln=%s  kernels %d  sum of kernel durations %.1f ms  clock %.2f GHz  MFMA busy %.3f" % (sys.argv[2], len(vs), dur / 1e6, g / dur, b / (g * 1024)))
PY
done
for c in 0 1 0 1; do CAPAMD_BERT_FUSED_LN=$c timeout 300 python $R/bench.py --steps 3 --warmup 1 --model bert --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('fused_ln=$c', round(r['value'],1), 'docs/s')"; done
