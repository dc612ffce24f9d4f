#!/bin/bash
# PMC pass for the GEMM kernels via scripts/gemm_bench.py; summary to gpurun_out/pmc_gemm_<tag>.txt
TAG=${1:-base}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pg_$TAG/1 -o p -- python $R/scripts/gemm_bench.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM --output-format csv -d /tmp/pg_$TAG/2 -o p -- python $R/scripts/gemm_bench.py > /dev/null 2>&1
python3 - <<PY > $R/gpurun_out/pmc_gemm_$TAG.txt
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pg_$TAG/*/*counter_collection.csv")+glob.glob("/tmp/pg_$TAG/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            key=r["Kernel_Name"].split("<")[1].split(">")[0]+" grid="+r["Grid_Size"]
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[key]["_dur_ns"].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
            acc[key]["_vgpr"].append(float(r["VGPR_Count"])); acc[key]["_lds"].append(float(r["LDS_Block_Size"]))
for key,d in sorted(acc.items()):
    print("==",key)
    for k,v in sorted(d.items()): print(f"   {k:32s} {sum(v)/len(v):16.1f} (n={len(v)})")
PY
cat $R/gpurun_out/pmc_gemm_$TAG.txt
