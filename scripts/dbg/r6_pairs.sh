#!/bin/bash
# round 6: lists_sims2_kernel (two lists per workgroup) against lists_sims_kernel (the nopairs build): parity tests of the list routes, then A/B
# (scripts/build_variant_obj.sh lists nopairs -DCAPAMD_LISTS_SIMS_PAIRS=0 first)
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "list" 2>&1 | tail -4
for r in 1 2; do
  for model in knrm drmm; do
    for lib in "" nopairs; do
      for S in 1 0; do
        libenv="X=1"; [ -n "$lib" ] && libenv="CAPAMD_LIB_PATH=$GRAFT_REPO_ROOT/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
        v=$(env $libenv timeout 600 python bench.py --model $model --steps 20 --warmup 4 --repeats 3 --step-streams $S --no-also --no-cpu-baseline --no-pmc-traffic --no-roofline-leg --no-pass-times 2>gpurun_out/ss_err.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f M  %.4f ms  (min %.4f max %.4f)' % (r['value']/1e6, r['ms_per_step'], r['repeats']['ms_per_step_min'], r['repeats']['ms_per_step_max']))" 2>&1 | tail -1)
        echo "$model lib=${lib:-pairs} step-streams=$S $v"
      done
    done
  done
done 2>&1 | tee gpurun_out/pairs_ab.txt
for q in 8 6; do
for lib in "" nopairs; do
  libenv="X=1"; [ -n "$lib" ] && libenv="CAPAMD_LIB_PATH=$GRAFT_REPO_ROOT/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
  v=$(env $libenv timeout 600 python bench.py --model knrm --qlen $q --steps 20 --warmup 4 --repeats 3 --no-also --no-cpu-baseline --no-pmc-traffic --no-roofline-leg --no-pass-times 2>gpurun_out/ss_err.txt | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f M  %.4f ms' % (r['value']/1e6, r['ms_per_step']))" 2>&1 | tail -1)
  echo "knrm qlen=$q lib=${lib:-new} $v"
done; done 2>&1 | tee -a gpurun_out/pairs_ab.txt
cd /tmp; export TMPDIR=/tmp
for lib in "" nopairs; do
  libenv="X=1"; [ -n "$lib" ] && libenv="CAPAMD_LIB_PATH=$GRAFT_REPO_ROOT/capreolus_amd/csrc/ablate/libcapreolus_amd_$lib.so"
  rm -rf /tmp/p; env $libenv timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline-leg --no-pass-times --step-streams 1 > /dev/null 2>&1
  echo "== kernel averages (us), serial steps, lib=${lib:-pairs}"
  python - <<PY
import csv,glob
f=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "lists_" in r["Name"]: print("   %-50s n=%s avg %.1f" % (r["Name"].split("::")[-1][:50], r["Calls"], float(r["AverageNs"])/1e3))
PY
done 2>&1 | tee -a $GRAFT_REPO_ROOT/gpurun_out/pairs_ab.txt
