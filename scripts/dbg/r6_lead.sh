#!/bin/bash
# round 6: the sims pass's leader list (CAPAMD_LISTS_SIMS_LEAD rows of the grid ahead): parity tests, A/B of lead distances, kernel averages
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "list" 2>&1 | tail -2
for r in 1 2; do
  for model in knrm drmm; do
    for lib in "" lead0 lead2 lead12 lead24; do
      libenv="X=1"; [ -n "$lib" ] && libenv="CAPAMD_LIB_PATH=$GRAFT_REPO_ROOT/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
      v=$(env $libenv timeout 600 python bench.py --model $model --steps 20 --warmup 4 --repeats 3 --step-streams 1 --no-also --no-cpu-baseline --no-pmc-traffic --no-roofline-leg --no-pass-times 2>gpurun_out/ss_err.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f M  %.4f ms  (min %.4f max %.4f)' % (r['value']/1e6, r['ms_per_step'], r['repeats']['ms_per_step_min'], r['repeats']['ms_per_step_max']))" 2>&1 | tail -1)
      echo "$model lib=${lib:-lead6} $v"
    done
  done
done 2>&1 | tee gpurun_out/lead_ab.txt
cd /tmp; export TMPDIR=/tmp
for lib in "" lead0 lead2 lead12 lead24; do
  libenv="X=1"; [ -n "$lib" ] && libenv="CAPAMD_LIB_PATH=$GRAFT_REPO_ROOT/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
  rm -rf /tmp/p; env $libenv timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline-leg --no-pass-times --step-streams 1 > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "lists_sims" in r["Name"]: print("lib=${lib:-lead6}   %-30s n=%s avg %.1f us" % (r["Name"].split("::")[1][:30], r["Calls"], float(r["AverageNs"])/1e3))
PY
done 2>&1 | tee -a $GRAFT_REPO_ROOT/gpurun_out/lead_ab.txt
