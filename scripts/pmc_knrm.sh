#!/bin/bash
# PMC passes for the KNRM kernel (variant from $1, extra bench args from $2); summaries to gpurun_out/pmc_<tag>.txt
V=${1:-0}; EXTRA=${2:-}; TAG=${3:-v$V}
export TMPDIR=/tmp CAPAMD_KNRM_VARIANT=$V
R=$GRAFT_REPO_ROOT; cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_READ_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$TAG/$i -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2>&1
done
python3 - <<PY > $R/gpurun_out/pmc_$TAG.txt
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_$TAG/*/*counter_collection.csv")+glob.glob("/tmp/pmc_$TAG/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "forward_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["_dur_ns"].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
for k,v in sorted(acc.items()): print(f"{k:45s} {sum(v)/len(v):18.1f}  (n={len(v)})")
PY
cat $R/gpurun_out/pmc_$TAG.txt
