#!/bin/bash
# round 6: library variants side by side on the list route (serial steps and the default's step streams), then the sims pass's kernel average and
# its fabric reads (FETCH_SIZE) per variant:   scripts/dbg/r6_libs_ab.sh default nopairs split1 ...   (names of capreolus_amd/csrc/ablate/ builds)
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
lp() { [ "$1" = default ] && echo "X=1" || echo "CAPAMD_LIB_PATH=$GRAFT_REPO_ROOT/capreolus_amd/csrc/ablate/libcapreolus_amd_$1.so"; }
for r in 1 2; do
  for model in knrm drmm; do
    for lib in "$@"; do
      for S in 1 0; do
        v=$(env $(lp $lib) timeout 600 python bench.py --model $model --steps 20 --warmup 4 --repeats 3 --step-streams $S --no-also --no-cpu-baseline --no-pmc-traffic --no-roofline-leg --no-pass-times 2>gpurun_out/ss_err.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f M  %.4f ms  (min %.4f max %.4f)' % (r['value']/1e6, r['ms_per_step'], r['repeats']['ms_per_step_min'], r['repeats']['ms_per_step_max']))" 2>&1 | tail -1)
        echo "$model lib=$lib step-streams=$S $v"
      done
    done
  done
done 2>&1 | tee gpurun_out/libs_ab.txt
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/p; env $(lp $lib) timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline-leg --no-pass-times --step-streams 1 > /dev/null 2>&1
  rm -rf /tmp/q; env $(lp $lib) timeout 400 rocprofv3 --output-format csv --pmc FETCH_SIZE -d /tmp/q -o c -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline-leg --no-pass-times --step-streams 1 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "lists_sims" in r["Name"]: print("lib=$lib   %-30s n=%s avg %.1f us" % (r["Name"].split("::")[1][:30], r["Calls"], float(r["AverageNs"])/1e3))
f=glob.glob("/tmp/q/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])) if f else []:
    if "lists_" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("lists_")[1].split("(")[0].split("<")[0]].append(float(r["Counter_Value"]))
print("lib=$lib   FETCH_SIZE KB per launch:", {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
done 2>&1 | tee -a $GRAFT_REPO_ROOT/gpurun_out/libs_ab.txt
