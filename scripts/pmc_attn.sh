#!/bin/bash
# PMC pass over the attention kernel of one BERT forward (profiling only)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pa; timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pa -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --model bert --docs 256 > /dev/null 2>&1
  f=$(find /tmp/pa -name "*counter_collection.csv")
  python - "$f" <<'PY'
import csv, collections, sys
agg = collections.defaultdict(float); n = 0; dur = 0
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    if 'attention' not in r['Kernel_Name']: continue
    agg[r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); dur += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print({k: "%.4g" % (v / max(1, len(seen))) for k, v in agg.items()}, "dispatches", len(seen), "avg us %.1f" % (dur / max(1, len(seen)) / 1e3))
PY
done
