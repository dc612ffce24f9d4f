#!/bin/bash
# Copies what scripts/gpu_round.sh left under gpurun_out/ (scratch) into profiles/<round>/ (tracked).  Usage: collect_profiles.sh r04
set -eu
D=profiles/${1:-r06}
G=gpurun_out
mkdir -p $D
for f in default knrm_serial_steps knrm_b1000 knrm_b1000_serial drmm_b1000 bert bert_skip_padding bert_fp16 bert_one_stream bert_r4mix drmmtks pacrr convknrm cedrknrm cedrknrm_separate_layernorm; do cp $G/bench_$f.json $D/; [ -f $G/bench_full_$f.json ] && cp $G/bench_full_$f.json $D/; done
for m in knrm knrm_step_streams knrm_roofline_leg drmm drmm_roofline_leg bert default drmmtks pacrr convknrm cedrknrm train_ConvKNRM train_PACRR; do
  f=$(ls $G/prof/$m/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $D/${m}_bench_kernel_stats.csv
done
for m in knrm knrm_roofline_leg drmm drmm_roofline_leg; do for c in fetch write tcc; do
  f=$(ls $G/prof/${m}_$c/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $D/${m}_${c}_counters.csv
done; done
for m in bert_mfma bert_mfma_r4mix; do f=$(ls $G/prof/$m/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 - "$f" "$D/${m}_counters_summary.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    f.write("kernel,launches,counter,mean_per_launch\n")
    for k, cs in acc.items():
        for c, v in cs.items():
            if len(v) >= 12:
                f.write(f'"{k}",{len(v)},{c},{sum(v) / len(v):.1f}\n')
PY
done
cp $G/knrm_hbm_traffic.json $G/drmm_hbm_traffic.json $D/
cp $G/pmc_summary.txt $D/pmc_summary.txt
cp $G/mfma_power.txt $D/mfma_power.txt
[ -f $G/hbm_read.txt ] && cp $G/hbm_read.txt $D/hbm_read.txt
[ -f $G/valu_rates.txt ] && cp $G/valu_rates.txt $D/valu_rates.txt
cp $G/pytest_gpu.log $D/pytest_gpu.log
[ -f $G/train_steps.jsonl ] && cp $G/train_steps.jsonl $D/train_steps.jsonl
[ -f $G/predict_e2e.json ] && cp $G/predict_e2e.json $D/predict_e2e.json
[ -f $G/knrm_pipes.txt ] && cp $G/knrm_pipes.txt $D/knrm_pipes.txt
for f in pmc_lists_knrm.txt pmc_lists_drmm.txt lists_ab.txt; do [ -f $G/$f ] && cp $G/$f $D/$f; done
ls -la $D | tail -40
