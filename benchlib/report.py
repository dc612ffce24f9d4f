"""The record -> bench_full.json + stderr, and ONE compact JSON line (<= 4 KB) as the last line of stdout."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # the repository root (bench.py lives there)


COMPACT_LIMIT = 4000      # bytes of the ONE stdout line (BENCH_r03: the driver could not parse a 22 KB line)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[: n - 3] + "..."


def _round(x):
    """floats to 6 significant digits (the stdout line only; bench_full.json keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.6g}")
    if isinstance(x, dict):
        return {k: _round(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_round(v) for v in x]
    return x


def compact_roofline(r, depth=0):
    """The roofline object of the stdout line: the contract's keys, the per-pass table of the list route and the per-pair HBM-bound leg,
    each cut down to numbers + kernel names (the notes / definitions stay in bench_full.json)."""
    if not isinstance(r, dict):
        return r
    out = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "compulsory_bytes", "traffic_over_compulsory", "traffic_over_requested", "kernel_ms",
                    "call_ms", "whole_step_frac", "whole_step_frac_nominal", "device_ms_per_step"))
    out.setdefault("traffic", r.get("traffic"))
    if "kernel" in r:
        out["kernel"] = _short(r["kernel"], 110 if depth == 0 else 70)
    if isinstance(r.get("passes"), list):
        out["passes"] = [{**_pick(q, ("ms", "bound", "frac", "frac_of_gather_only_ceiling")), "kernel": _short(q.get("kernel", ""), 44)} for q in r["passes"]]
    if isinstance(r.get("per_pair_hbm_leg"), dict) and depth == 0:
        out["per_pair_hbm_leg"] = compact_roofline(r["per_pair_hbm_leg"], 1)
    return out


def compact(rec, full_path):
    """What the driver parses: the contract's keys + config + roofline + cpu_baseline of the headline, and one short entry per `also` leg."""
    out = _pick(rec, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"))
    out["vs_baseline"] = rec.get("vs_baseline")
    if isinstance(rec.get("repeats"), dict):
        out["repeats"] = _pick(rec["repeats"], ("n", "ms_per_step_min", "ms_per_step_max", "value_min", "value_max"))
    cfg = rec.get("config", {})
    out["config"] = {**_pick(cfg, ("pairs_per_step_per_gpu", "passages_per_s", "streams", "parallelism")), "workload": _short(cfg.get("workload", ""), 330)}
    out["roofline"] = compact_roofline(rec.get("roofline"))
    cb = rec.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {**_pick(cb, ("value", "unit", "cores", "kind", "aten_port_value", "aten_port_threads", "aten_port_batch", "aten_port_reference_default",
                                              "config0_s", "config0_gpu_s")), "sample": _short(cb.get("sample", ""), 90)}
    if isinstance(rec.get("collective"), dict):
        out["collective"] = _pick(rec["collective"], ("backend", "rccl_ranks", "gathered_bytes_per_step", "gather_ms"))
    if isinstance(rec.get("oracle_check"), dict):
        out["oracle_check"] = rec["oracle_check"]
    if isinstance(rec.get("parity"), dict):
        out["parity"] = _pick(rec["parity"], ("dtype", "documents", "max_score_error_of_scale_vs_fp32_port"))
    if isinstance(rec.get("other_operand_type"), dict):
        out["other_operand_type"] = _pick(rec["other_operand_type"], ("dtype", "value", "ms_per_step", "whole_step_frac", "whole_step_frac_nominal"))
    if isinstance(rec.get("zero_idf_run"), dict):
        out["zero_idf_run"] = _pick(rec["zero_idf_run"], ("value", "unit", "ms_per_step", "steps"))
    if isinstance(rec.get("step_streams"), dict):
        out["step_streams"] = {"streams": rec["step_streams"].get("streams"),
                               "serial_steps": _pick(rec["step_streams"].get("serial_steps") or {}, ("value", "ms_per_step", "call_ms"))}
    if isinstance(rec.get("resident_int32_route"), dict):
        out["resident_int32_route"] = _pick(rec["resident_int32_route"], ("value", "unit", "ms_per_step", "error"))
    if isinstance(rec.get("qlen8_lists"), dict):
        out["qlen8_lists"] = _pick(rec["qlen8_lists"], ("value", "ms_per_step", "oracle_check_max_err_of_scale", "error"))
    if isinstance(rec.get("lists_on_uniform_ids"), dict):
        out["lists_on_uniform_ids"] = _pick(rec["lists_on_uniform_ids"], ("value", "ms_per_step", "mean_distinct_terms_per_list", "error"))
    also = []
    for a in rec.get("also", []) or []:
        if not isinstance(a, dict) or "error" in a:
            also.append(a if isinstance(a, dict) else {"error": str(a)})
            continue
        e = {"workload": " ".join(str(a.get("config", {}).get("workload", "")).split()[:2]), **_pick(a, ("value", "unit", "ms_per_step", "steps", "dtype"))}
        r = a.get("roofline") or {}
        e["roofline"] = {**_pick(r, ("bound", "frac", "whole_step_frac", "whole_step_frac_nominal")), "kernel": _short(r.get("kernel", ""), 40)}
        if isinstance(r.get("per_pair_hbm_leg"), dict):
            e["roofline"]["per_pair_hbm_leg_frac"] = r["per_pair_hbm_leg"].get("frac")
        if isinstance(a.get("cpu_baseline"), dict):
            e["cpu_baseline"] = _pick(a["cpu_baseline"], ("value", "cores", "kind"))
        if isinstance(a.get("oracle_check"), dict):
            e["oracle_err"] = a["oracle_check"].get("max_err_of_scale")
        if isinstance(a.get("other_operand_type"), dict):
            e["other_operand_type"] = _pick(a["other_operand_type"], ("dtype", "value", "whole_step_frac", "whole_step_frac_nominal"))
        if isinstance(a.get("zero_idf_run"), dict):
            e["zero_idf_run"] = _pick(a["zero_idf_run"], ("value", "ms_per_step"))
        if isinstance(a.get("step_streams"), dict):
            e["step_streams"] = a["step_streams"].get("streams")
        also.append(e)
    if also:
        out["also"] = also
    out["full_record"] = full_path
    out = _round(out)
    line = json.dumps(out, separators=(",", ":"))
    for victim in ("also", "other_operand_type", "parity"):        # (never reached with the default legs: a guard, not a plan)
        if len(line) <= COMPACT_LIMIT:
            break
        if victim == "also" and "also" in out:
            out["also"] = [{k: v for k, v in e.items() if k in ("workload", "value", "ms_per_step", "error")} for e in out["also"]]
        else:
            out.pop(victim, None)
        line = json.dumps(out, separators=(",", ":"))
    return line


def emit(rec):
    """The whole record goes to bench_full.json (next to this file; `also` legs in full, CPU sweeps, per-pass work figures, notes) and to
    stderr; stdout gets ONE compact JSON line (<= 4 KB) - the last line of stdout: RCCL prints a version banner through C stdio, which
    would otherwise be flushed at process exit, after Python's own line."""
    full_path = os.path.join(ROOT, "bench_full.json")
    try:
        with open(full_path, "w") as f:
            json.dump(rec, f, indent=1)
        shown = "bench_full.json"
    except OSError as e:
        shown = f"not written ({type(e).__name__})"
    print(json.dumps(rec), file=sys.stderr, flush=True)
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(compact(rec, shown), flush=True)

