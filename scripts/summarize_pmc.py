"""Turns the rocprofv3 --pmc CSVs of scripts/gpu_round.sh into the per-launch figures quoted in DESIGN.md and profiles/<round>/*.json:
HBM-side traffic of the KNRM / DRMM kernels (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md section HBM), L2 hit rates, and the MFMA
duty cycle of the BERT GEMM kernels."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]


def counters(sub, match):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if any(m in r["Kernel_Name"] for m in match):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def counters_per_call(sub, part, whole):
    """For a call that is several kernels (the whole-list route): counter sums over the kernels whose name contains `part`, per launch of
    the kernel whose name contains `whole` (the pooling pass: one per call)."""
    tot, calls = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(os.path.join(root, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if part in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                if whole in r["Kernel_Name"]:
                    calls[r["Counter_Name"]] += 1
    return {k: (tot[k] / calls[k], calls[k]) for k in tot if calls.get(k)}


out = {}
for model in ("knrm", "drmm"):
    rec = {}
    for leg in ("", "_roofline_leg"):
        name = model + leg
        f = counters(name + "_fetch", ("forward_kernel", "stream_kernel")).get("FETCH_SIZE")
        w = counters(name + "_write", ("forward_kernel", "stream_kernel")).get("WRITE_SIZE")
        t = counters(name + "_tcc", ("forward_kernel", "stream_kernel"))
        route = "per-pair kernel"
        if not f or not w:      # the headline leg scored as whole candidate lists: mark + query + sims + pool per call
            f = counters_per_call(name + "_fetch", "lists_", "_pool_").get("FETCH_SIZE")
            w = counters_per_call(name + "_write", "lists_", "_pool_").get("WRITE_SIZE")
            t = counters_per_call(name + "_tcc", "lists_", "_pool_")
            route = "whole candidate lists: sums over lists_mark / lists_query / lists_sims / lists_*_pool per call"
        if not f or not w:
            continue
        hit, miss = t.get("TCC_HIT_sum", (0, 0))[0], t.get("TCC_MISS_sum", (0, 0))[0]
        rec["headline_leg" if not leg else "roofline_leg"] = {
            "route": route, "launches_sampled": f[1], "FETCH_SIZE_KB_per_launch": f[0], "WRITE_SIZE_KB_per_launch": w[0],
            "hbm_bytes_per_launch": f[0] * 1024 * 2 + w[0] * 1024, "l2_hit_rate": hit / (hit + miss) if hit + miss else None}
    rec["correction"] = ("MI355X_MICROARCH.md section HBM: gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide (16 B/lane) coalesced reads -> x2; "
                         "bytes = KB*1024; memory-side (fabric) requests of the L2s - Infinity-Cache hits are counted, so this bounds HBM traffic from above")
    rec["command"] = "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum (separate passes) -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also --no-roofline-leg [--uniform-ids --vocab 4000001 --batches 2]"
    out[model] = rec
    json.dump(rec, open(os.path.join(os.path.dirname(root.rstrip("/")), f"{model}_hbm_traffic.json"), "w"), indent=1)
    print(model, json.dumps(rec, indent=1)[:1500])
for sub in ("bert_mfma", "bert_mfma_pingpong"):
    for kern in ("gemm_ring_kernelILi1", "gemm_ring_kernelILi3", "gemm_ring_kernelILi5", "gemm_pingpong_kernelILi1", "gemm_pingpong_kernelILi3", "gemm_pingpong_kernelILi5",
                 "attention_persistent", "attention_s256"):
        c = counters(sub, (kern,))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
            busy, active = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / 1024.0, c["GRBM_GUI_ACTIVE"][0] / 8.0
            print(f"{sub:20s} {kern:28s} n={c['GRBM_GUI_ACTIVE'][1]:4d}  MFMA pipe busy {busy:10.0f} of {active:10.0f} cycles per launch = {busy / active:.3f}")
