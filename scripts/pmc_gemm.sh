#!/bin/bash
# PMC passes for the GEMM kernels via scripts/gemm_bench.py; summary to gpurun_out/pmc_gemm_<tag>.txt
TAG=${1:-base}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "FETCH_SIZE" "WRITE_SIZE" \
         "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pg_$TAG/$i -o p -- python $R/scripts/gemm_bench.py > /tmp/pg_$TAG.$i.log 2>&1
done
python3 - <<PY > $R/gpurun_out/pmc_gemm_$TAG.txt
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pg_$TAG/*/*counter_collection.csv")+glob.glob("/tmp/pg_$TAG/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            key=r["Kernel_Name"].split("<")[1].split(">")[0]
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[key]["_dur_ns"].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
for key,d in sorted(acc.items()):
    print("==",key)
    for k,v in sorted(d.items()): print(f"   {k:36s} {sum(v)/len(v):18.1f} (n={len(v)})")
PY
cat $R/gpurun_out/pmc_gemm_$TAG.txt
