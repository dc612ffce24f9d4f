#!/bin/bash
# PMC passes (counters only, one group per run) of the KNRM headline leg: scripts/dbg/pmc_stream.sh name[:ENV=V,...][@libname] ...
set -u
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also --no-roofline-leg --no-pmc-traffic"
for spec in "$@"; do
  name=${spec%%[:@]*}; envs=""; lib=""
  case "$spec" in *@*) lib=${spec##*@};; esac
  case "$spec" in *:*) envs=${spec#*:}; envs=${envs%%@*}; envs=${envs//,/ };; esac
  libenv=""; [ -n "$lib" ] && libenv="CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
  echo "== $name"
  for grp in "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA_RDREQ_sum" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
    rm -rf /tmp/pk; env $envs $libenv CAPAMD_BENCH_NO_CHECK=1 timeout 300 rocprofv3 --output-format csv --pmc $grp -d /tmp/pk -o c -- $B > /dev/null 2>&1
    python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pk/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])) if f else []:
    if "knrm_forward" in r["Kernel_Name"] or "stream_kernel" in r["Kernel_Name"] or "drmm_forward" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  ", {k: round(sum(v) / len(v)) for k, v in acc.items()}, "launches", max([len(v) for v in acc.values()] or [0]))
PY
  done
done 2>&1 | tee $R/gpurun_out/pmc_stream.txt
