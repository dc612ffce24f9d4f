export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
cd /tmp
run() { # name, counters
  rm -rf /tmp/pm_$1
  timeout 200 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pm_$1 -o x -- python $GRAFT_REPO_ROOT/scripts/dbg/attn_ab.py /tmp/o.pt 256 > /dev/null 2>&1
  python - "$1" <<PY
import csv,glob,sys,collections
f=glob.glob("/tmp/pm_%s/**/*counter_collection.csv"%sys.argv[1],recursive=True)
acc=collections.defaultdict(float)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "attention" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
print(sys.argv[1], dict(acc))
PY
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run c "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
run d "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"
run e "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES"
