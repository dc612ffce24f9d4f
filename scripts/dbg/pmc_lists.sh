#!/bin/bash
# PMC passes (counters only, one group per run) of the lists route's kernels: scripts/dbg/pmc_lists.sh model name[:ENV=V,...][@libname] ...
set -u
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
model=$1; shift
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also --no-roofline-leg --no-pmc-traffic --step-streams 1 --model $model"
for spec in "$@"; do
  name=${spec%%[:@]*}; envs=""; lib=""
  case "$spec" in *@*) lib=${spec##*@};; esac
  case "$spec" in *:*) envs=${spec#*:}; envs=${envs%%@*}; envs=${envs//,/ };; esac
  libenv="X=1"; [ -n "$lib" ] && libenv="CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
  echo "== $model $name"
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE TCP_TCC_READ_REQ_sum" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
    rm -rf /tmp/pk; env $envs $libenv CAPAMD_BENCH_NO_CHECK=1 timeout 300 rocprofv3 --output-format csv --pmc $grp -d /tmp/pk -o c -- $B > /dev/null 2>&1
    python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pk/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])) if f else []:
    if "lists_" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("lists_")[1].split("(")[0].split("<")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("  %-18s" % k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
  done
done 2>&1 | tee $R/gpurun_out/pmc_lists_$model.txt
