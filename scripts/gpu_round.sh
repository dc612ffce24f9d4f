#!/bin/bash
# One GPU-box session (round 6): parity tests, smoke, the bench lines, rocprofv3 kernel stats + PMC traffic.  Outputs under gpurun_out/
# (scripts/collect_profiles.sh r06 copies what is kept into profiles/r06/).  bench.py prints ONE compact line and writes the whole record
# to bench_full.json: both are kept per run (bench_<name>.json = the line, bench_full_<name>.json = the record).
set -u
rm -rf gpurun_out/prof gpurun_out/ab_*.json; mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py"
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 ) > gpurun_out/pytest_gpu.log 2>&1; cat gpurun_out/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1
# ---- bench lines -------------------------------------------------------------------------------------------------------
timeout 600 $B --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_default.json; cp bench_full.json gpurun_out/bench_full_default.json 2>/dev/null
timeout 300 $B --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline-leg --step-streams 1 2>/dev/null | tail -1 > gpurun_out/bench_knrm_serial_steps.json; cp bench_full.json gpurun_out/bench_full_knrm_serial_steps.json 2>/dev/null
timeout 300 $B --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc-traffic --launch-docs 1000 2>/dev/null | tail -1 > gpurun_out/bench_knrm_b1000.json; cp bench_full.json gpurun_out/bench_full_knrm_b1000.json 2>/dev/null
timeout 300 $B --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pmc-traffic --launch-docs 1000 --launch-streams 1 --no-graph 2>/dev/null | tail -1 > gpurun_out/bench_knrm_b1000_serial.json; cp bench_full.json gpurun_out/bench_full_knrm_b1000_serial.json 2>/dev/null
timeout 300 $B --steps 20 --warmup 3 --no-cpu-baseline --no-pmc-traffic --model drmm --launch-docs 1000 2>/dev/null | tail -1 > gpurun_out/bench_drmm_b1000.json; cp bench_full.json gpurun_out/bench_full_drmm_b1000.json 2>/dev/null
timeout 600 $B --steps 5 --warmup 2 --model bert 2>/dev/null | tail -1 > gpurun_out/bench_bert.json; cp bench_full.json gpurun_out/bench_full_bert.json 2>/dev/null
timeout 600 $B --steps 5 --warmup 2 --model bert --no-cpu-baseline --no-bert-other-dtype --bert-skip-padding 2>/dev/null | tail -1 > gpurun_out/bench_bert_skip_padding.json; cp bench_full.json gpurun_out/bench_full_bert_skip_padding.json 2>/dev/null
timeout 600 $B --steps 5 --warmup 2 --model bert --no-cpu-baseline --no-bert-other-dtype --bert-dtype fp16 2>/dev/null | tail -1 > gpurun_out/bench_bert_fp16.json; cp bench_full.json gpurun_out/bench_full_bert_fp16.json 2>/dev/null
timeout 600 $B --steps 5 --warmup 2 --model bert --no-cpu-baseline --no-bert-other-dtype --bert-streams 1 2>/dev/null | tail -1 > gpurun_out/bench_bert_one_stream.json; cp bench_full.json gpurun_out/bench_full_bert_one_stream.json 2>/dev/null
CAPAMD_GEMM_PICK=qkv=128,ffn1=128,oproj=256x32,ffn2=256x32 timeout 600 $B --steps 5 --warmup 2 --model bert --no-cpu-baseline --no-bert-other-dtype 2>/dev/null | tail -1 > gpurun_out/bench_bert_r4mix.json; cp bench_full.json gpurun_out/bench_full_bert_r4mix.json 2>/dev/null
for mdl in drmmtks pacrr convknrm; do timeout 300 $B --steps 20 --warmup 3 --no-cpu-baseline --model $mdl 2>/dev/null | tail -1 > gpurun_out/bench_$mdl.json; cp bench_full.json gpurun_out/bench_full_$mdl.json 2>/dev/null; done
PYTHONPATH=$R timeout 300 python $R/scripts/sibling_bench.py --only CEDRKNRM 2>/dev/null | tail -1 > gpurun_out/bench_cedrknrm.json; cp bench_full.json gpurun_out/bench_full_cedrknrm.json 2>/dev/null; cat gpurun_out/bench_cedrknrm.json
PYTHONPATH=$R timeout 300 python $R/scripts/predict_e2e_bench.py 2>/dev/null | tail -1 > gpurun_out/predict_e2e.json; cat gpurun_out/predict_e2e.json
timeout 120 ./scripts/ubench/valu_rates > gpurun_out/valu_rates.txt 2>&1
PYTHONPATH=$R timeout 300 python $R/scripts/train_step_bench.py 2>/dev/null | grep "^{" > gpurun_out/train_steps.jsonl; cat gpurun_out/train_steps.jsonl
CAPAMD_CEDR_FUSED=0 PYTHONPATH=$R timeout 300 python $R/scripts/sibling_bench.py --only CEDRKNRM 2>/dev/null | tail -1 > gpurun_out/bench_cedrknrm_separate_layernorm.json; cp bench_full.json gpurun_out/bench_full_cedrknrm_separate_layernorm.json 2>/dev/null; cat gpurun_out/bench_cedrknrm_separate_layernorm.json
python - <<'PY'
import json
for f in ("default", "knrm_serial_steps", "knrm_b1000", "knrm_b1000_serial", "drmm_b1000", "bert", "bert_skip_padding", "bert_fp16", "bert_one_stream", "bert_r4mix", "drmmtks", "pacrr", "convknrm"):
    try:
        r = json.load(open(f"gpurun_out/bench_{f}.json")); ro = r["roofline"]
        print(f"{f:20s} {r['value']:14.1f} {r['unit']}  ms/step {r['ms_per_step']:.3f}  roofline frac {ro.get('frac')}  {ro.get('whole_step_frac_nominal', '')}")
        for leg in r.get("also", []):
            print(f"   also: {str(leg.get('config', {}).get('workload', leg))[:60]} -> {leg.get('value')} frac {leg.get('roofline', {}).get('frac')}")
    except Exception as e:
        print(f, "FAILED", e)
PY
# ---- rocprofv3 kernel stats of the same commands (one kernel of interest per run, so that the average is that kernel's) -------
cd /tmp; P=/tmp/prof; rm -rf $P; mkdir -p $P
KS="rocprofv3 --kernel-trace --stats --output-format csv"
timeout 300 $KS -d $P/knrm -o knrm -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-roofline-leg --step-streams 1 > /dev/null 2>&1      # (serial steps: a kernel's average is its own duration)
timeout 300 $KS -d $P/knrm_step_streams -o knrm -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-roofline-leg --no-pass-times > /dev/null 2>&1      # (the default: two step streams - kernels of consecutive steps overlap and stretch)
timeout 300 $KS -d $P/knrm_roofline_leg -o knrm -- $B --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-roofline-leg --uniform-ids --vocab 4000001 --batches 2 > /dev/null 2>&1
timeout 300 $KS -d $P/drmm -o drmm -- $B --steps 10 --warmup 2 --no-cpu-baseline --no-roofline-leg --model drmm > /dev/null 2>&1
timeout 300 $KS -d $P/drmm_roofline_leg -o drmm -- $B --steps 10 --warmup 2 --no-cpu-baseline --no-roofline-leg --model drmm --uniform-ids --vocab 4000001 --batches 2 > /dev/null 2>&1
timeout 300 $KS -d $P/bert -o bert -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-bert-other-dtype --bert-streams 1 --model bert > /dev/null 2>&1
timeout 300 $KS -d $P/default -o default -- $B --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
for mdl in drmmtks pacrr convknrm; do timeout 300 $KS -d $P/$mdl -o $mdl -- $B --steps 20 --warmup 3 --no-cpu-baseline --model $mdl > /dev/null 2>&1; done
PYTHONPATH=$R timeout 300 $KS -d $P/cedrknrm -o cedrknrm -- python $R/scripts/sibling_bench.py --only CEDRKNRM > /dev/null 2>&1
for mdl in ConvKNRM PACRR; do PYTHONPATH=$R timeout 300 $KS -d $P/train_$mdl -o t -- python $R/scripts/train_step_bench.py --only $mdl --steps 40 > /dev/null 2>&1; done
# ---- PMC passes (own runs, counters only): HBM traffic of the KNRM / DRMM headline and roofline legs, MFMA busy of the BERT GEMMs ------
PM="rocprofv3 --output-format csv --pmc"
for leg in "knrm:" "knrm_roofline_leg:--uniform-ids --vocab 4000001 --batches 2" "drmm:--model drmm" "drmm_roofline_leg:--model drmm --uniform-ids --vocab 4000001 --batches 2"; do
  name=${leg%%:*}; extra=${leg#*:}
  timeout 300 $PM FETCH_SIZE -d $P/${name}_fetch -o c -- $B --steps 5 --warmup 1 --no-cpu-baseline --no-also --no-roofline-leg $extra > /dev/null 2>&1
  timeout 300 $PM WRITE_SIZE -d $P/${name}_write -o c -- $B --steps 5 --warmup 1 --no-cpu-baseline --no-also --no-roofline-leg $extra > /dev/null 2>&1
  timeout 300 $PM TCC_HIT_sum TCC_MISS_sum -d $P/${name}_tcc -o c -- $B --steps 5 --warmup 1 --no-cpu-baseline --no-also --no-roofline-leg $extra > /dev/null 2>&1
done
timeout 300 $PM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d $P/bert_mfma -o c -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-bert-other-dtype --bert-streams 1 --model bert --docs 256 > /dev/null 2>&1
CAPAMD_GEMM_PICK=qkv=128,ffn1=128,oproj=256x32,ffn2=256x32 timeout 300 $PM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d $P/bert_mfma_r4mix -o c -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-bert-other-dtype --bert-streams 1 --model bert --docs 256 > /dev/null 2>&1
cd $R
# what travels back (gpurun_out/ is capped at 64 MiB): the kernel statistics and agent info as they are; of the counter files only the rows of this
# library's kernels; no kernel traces
python3 - $P gpurun_out/prof <<'PY'
import csv, os, shutil, sys
src, dst = sys.argv[1:3]
for root, _, files in os.walk(src):
    for f in files:
        if not f.endswith(".csv") or f.endswith("kernel_trace.csv"):
            continue
        rel = os.path.relpath(root, src).split(os.sep)[0]
        os.makedirs(os.path.join(dst, rel), exist_ok=True)
        out = os.path.join(dst, rel, f)
        if f.endswith("counter_collection.csv"):
            with open(os.path.join(root, f)) as fi, open(out, "w", newline="") as fo:
                r = csv.DictReader(fi)
                w = csv.DictWriter(fo, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value"])
                w.writeheader()
                for row in r:
                    k = row["Kernel_Name"]
                    if any(t in k for t in ("capamd", "lists_", "forward_kernel", "stream_kernel", "gemm_", "attention_")):
                        w.writerow({"Kernel_Name": k[:120], "Counter_Name": row["Counter_Name"], "Counter_Value": row["Counter_Value"]})
        elif os.path.getsize(os.path.join(root, f)) < 4000 * 1024:
            shutil.copy(os.path.join(root, f), out)
PY
python scripts/summarize_pmc.py gpurun_out/prof > gpurun_out/pmc_summary.txt 2>&1; cat gpurun_out/pmc_summary.txt
timeout 200 ./scripts/ubench/mfma_power > gpurun_out/mfma_power.txt 2>&1; cat gpurun_out/mfma_power.txt
[ -x ./scripts/ubench/hbm_read ] && { timeout 200 ./scripts/ubench/hbm_read > gpurun_out/hbm_read.txt 2>&1; cat gpurun_out/hbm_read.txt; }
ls gpurun_out/prof/
# which pipe binds the per-pair streaming kernel (bench.py --per-pair) and the kernels of the whole-list route: issue activity, wait buckets
cd $R; bash scripts/dbg/pmc_knrm_pipes.sh --per-pair > gpurun_out/knrm_pipes.txt 2>&1; cat gpurun_out/knrm_pipes.txt
bash scripts/dbg/pmc_lists.sh knrm base > /dev/null 2>&1; bash scripts/dbg/pmc_lists.sh drmm base > /dev/null 2>&1; cat gpurun_out/pmc_lists_knrm.txt gpurun_out/pmc_lists_drmm.txt
bash scripts/dbg/lists_variants.sh base pairs > gpurun_out/lists_ab.txt 2>&1; cat gpurun_out/lists_ab.txt
