"""Short legs of the row-N4 models (DRMM-TKS, PACRR, ConvKNRM) on candidate lists of the KNRM benchmark's shape."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

from benchlib.common import Ctx1, HBM_PEAK_GBS, SIMS_PIPE, _tables, algorithmic_bytes_per_pair, table, timed_loop
from benchlib.interaction import lists_roofline


def sibling_leg(args, ctx, model, V, uniform, steps, warmup, seed):
    """One timed leg of a row-N4 model on candidate lists of the KNRM benchmark's shape: (reranker module, batch, last scores, wall
    seconds, per-step device seconds, bytes of one gathered row, mean non-pad terms per document)."""
    from types import SimpleNamespace

    from capreolus_amd import engine, synthetic
    from capreolus_amd.reranker import DRMMTKS, PACRR, ConvKNRM

    world, dev, use_dist, dist = ctx.world, ctx.dev, ctx.use_dist, ctx.dist
    Q, L, D = 4, 800, args.dim
    n_queries = args.queries or 64
    n_pairs = n_queries * args.docs
    emb = table(dev, V, D)
    batch = synthetic.make_candidate_list_torch(n_queries, args.docs, V, dev, seed=seed, maxqlen=Q, maxdoclen=L, uniform_ids=uniform)
    if model == "convknrm":      # nn.Embedding ids only (the slowembedtext extractor has no negative OOV ids)
        batch = {k: (v.abs() if v.dtype == torch.int64 else v) for k, v in batch.items()}
    torch.manual_seed(0)
    stub = SimpleNamespace(embeddings=np.zeros((2, D), dtype=np.float32), config={"maxqlen": Q}, pad=0)
    rr = {"drmmtks": DRMMTKS, "pacrr": PACRR, "convknrm": ConvKNRM}[model]({}, stub)
    m = rr.build_model().to(dev).eval()
    name = "embeddings" if model == "convknrm" else "embedding"
    setattr(m, name, torch.nn.Embedding.from_pretrained(emb, freeze=True))
    q_all, d_all, idf_all = batch["query"], batch["posdoc"], batch["query_idf"]
    gathered = torch.empty(n_pairs * world, dtype=torch.float32, device=dev) if use_dist else None
    out = [None]

    # whole candidate lists (csrc/lists.hip) where the model takes them, unless asked otherwise or on the uniform-id leg (lists share nothing there)
    # (--force-lists: ConvKNRM's list entry, which exists and is bit-identical but is not the default route - 2 % slower than its per-pair kernel)
    as_lists = (bool(getattr(rr, "supports_lists", False)) or (getattr(args, "force_lists", False) and hasattr(m, "forward_lists"))) and \
        not args.per_pair and not uniform and n_queries >= 2
    offsets = np.arange(0, n_pairs + 1, args.docs, dtype=np.int64)

    # consecutive steps round-robin over two HIP streams where the step is a call of few lists (benchlib/interaction.py):
    # per-stream workspace (engine._lists_workspace) and score tensor
    ns = getattr(args, "step_streams", 1)
    if ns <= 0:
        ns = 2 if n_queries <= 128 else 1
    side = [torch.cuda.Stream(device=dev) for _ in range(ns)] if (ns > 1 and as_lists and not use_dist) else []

    def score():
        return m.forward_lists(offsets, query=q_all, doc=d_all, idf=idf_all).view(-1) if as_lists else m(d_all, q_all, idf_all).view(-1)

    def step(i):
        with torch.no_grad():
            if side:
                with torch.cuda.stream(side[i % len(side)]):
                    out[0] = score()
            else:
                out[0] = score()
        if use_dist:
            dist.all_gather_into_tensor(gathered, out[0])

    def drain():
        for st in side:
            torch.cuda.current_stream().wait_stream(st)

    with torch.no_grad():
        m(d_all[:8], q_all[:8], idf_all[:8])          # packs the tables and checks the status word once, synchronously
    status = engine.deferred_status(dev)              # the timed calls are queued back to back like the KNRM / DRMM launches
    status.__enter__()                                # (check=False there); the accumulated status bits are raised at the end
    for k in range(len(side)):                        # (every stream's workspace and module load outside the timed region)
        step(k)
    drain()
    elapsed, kern_s = timed_loop(ctx, step, warmup, steps, drain if side else None)   # one scoring call = the model's kernel + a few tiny torch ops of the mirror
    torch.cuda.synchronize()
    assert os.environ.get("CAPAMD_BENCH_NOCHECK") == "1" or torch.isfinite(out[0]).all()   # (the knob: profiling builds that drop a phase of the kernel)
    status.__exit__(None, None, None)
    nonpad = float((d_all > 0).sum().item()) / n_pairs
    passes = None
    if as_lists and not args.no_pass_times:       # the route's passes, HIP events between them (the -DCAPAMD_PROFILING build of the same kernels)
        from capreolus_amd import _lib

        with _lib.profiling_build() as lib, torch.no_grad():
            step(0)
            torch.cuda.synchronize()
            lib.capamd_debug_lists_timing(1)
            try:
                for _ in range(3):
                    step(0)
                ms = (ctypes.c_double * 8)()
                groups = lib.capamd_debug_lists_timing_read(ms)
            finally:
                lib.capamd_debug_lists_timing(0)
            torch.cuda.synchronize()
        passes = [x / 3 for x in ms][:5] if groups else None
    if model == "convknrm":
        G, F = m.p["maxngram"], m.p["filters"]
        row = G * (G + 1) // 2 * F * 4
    else:
        row = 4 * (m._packed.get(getattr(m, name).weight).numel() // V)
    return rr, m, batch, out[0], elapsed, kern_s, row, nonpad, as_lists, passes, len(side) or 1


def sibling_oracle(model, m, D, q, d, idf, emb_h):
    """The C oracle's scorer of a row-N4 model on host arrays (checker of the timed scores and the `cpu_baseline` port)."""
    from oracle import cpu as oracle

    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if "embedding" not in k}
    if model == "drmmtks":
        packed = oracle.pack(emb_h)
        return lambda: oracle.drmmtks(q, d, idf, packed, D, m.topk, sd["gates.weight"], sd["ffw.0.weight"], sd["ffw.0.bias"],
                                      sd["output_layer.weight"], sd["output_layer.bias"])
    if model == "pacrr":
        packed = oracle.pack(emb_h)
        p = m.p
        n_ng = p["maxgram"] - p["mingram"] + 1
        return lambda: oracle.pacrr(q, d, idf, packed, D, p["mingram"], p["maxgram"], p["nfilters"], p["kmax"],
                                    [sd[f"ngrams.{i}.conv.weight"] for i in range(n_ng)], [sd[f"ngrams.{i}.conv.bias"] for i in range(n_ng)], p["idf"],
                                    sd["linear1.weight"], sd["linear1.bias"], sd["linear2.weight"], sd["linear2.bias"], sd["linear3.weight"],
                                    sd["linear3.bias"], p["nonlinearity"])
    p = m.p
    mu, sigma = (x.cpu().numpy() for x in m.kernels.stacked())
    return lambda: oracle.convknrm(q, d, emb_h, [sd[f"convs.{i}.0.weight"] for i in range(p["maxngram"])],
                                   [sd[f"convs.{i}.0.bias"] for i in range(p["maxngram"])], p["crossmatch"], mu, sigma, sd["combine.0.weight"],
                                   sd["combine.0.bias"])


def bench_sibling(args, ctx, model=None, steps=None, warmup=None, with_cpu=None, check_pairs=256):
    """Row N4 models on the KNRM benchmark's candidate lists: DRMM-TKS, PACRR (KNRM's gather; same algorithmic bytes) and
    ConvKNRM (per position 6 projection-table parts of `filters` floats instead of one embedding row).  The timed
    scores of the first `check_pairs` pairs are checked against the C oracle in every run; `roofline.frac` comes from a second leg
    on uniform ids over the `--roofline-vocab` table (where HBM binds), like the KNRM / DRMM lines."""
    model = model or args.model
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    with_cpu = (not args.no_cpu_baseline) if with_cpu is None else with_cpu
    world, rank, dev = ctx.world, ctx.rank, ctx.dev
    Q, L, V, D = 4, 800, args.vocab, args.dim
    n_queries = args.queries or 64
    n_pairs = n_queries * args.docs
    rr, m, batch, scores, elapsed, kern_s, row, nonpad, as_lists, passes, n_streams = sibling_leg(args, ctx, model, V, args.uniform_ids, steps, warmup, 1 + rank)
    q_all, d_all, idf_all = batch["query"], batch["posdoc"], batch["query_idf"]
    emb = table(dev, V, D)
    if model == "convknrm":
        abytes = L * (8 + row) + Q * (8 + row) + 4
        kname = "convknrm_forward_kernel<2>"
    else:
        abytes = algorithmic_bytes_per_pair("knrm", Q, L, D) + 4 * Q
        kname = {"drmmtks": "drmmtks_forward_kernel<5, 12>", "pacrr": "pacrr_mfma_kernel<5, 2>"}[model]
    headline_kernel = ("lists_mark_kernel + lists_query_kernel<5> + lists_sims_kernel<5, false> + "
                       + {"drmmtks": "lists_tks_pool_kernel<12>", "pacrr": "pacrr_mfma_lists_kernel<5, 2>"}.get(model, "")) if as_lists else kname
    requested = n_pairs * (L * 8 + Q * 8 + (nonpad + Q) * row + 4) / kern_s / 1e9
    if rank != 0:
        return None
    # the timed scores are the oracle's (a bounded sample; oracle/ is the checker here, never the thing measured)
    emb_h = emb.cpu().numpy() if V <= 400001 else None
    oracle_err = None
    if emb_h is not None and os.environ.get("CAPAMD_BENCH_NOCHECK") != "1":
        nchk = min(check_pairs, n_pairs)
        want, err = sibling_oracle(model, m, D, *(t[:nchk].cpu().numpy() for t in (q_all, d_all, idf_all)), emb_h)()
        assert err == 0
        oracle_err = float(np.abs(scores[:nchk].cpu().numpy() - want).max() / max(1.0, np.abs(want).max()))
        assert oracle_err <= 1e-3, f"{model}: the timed scores differ from the oracle's by {oracle_err}"
    hbm = None
    if not args.no_roofline_leg and world == 1 and not args.uniform_ids and args.roofline_vocab > V:
        del batch, scores
        big = sibling_leg(args, Ctx1(ctx), model, args.roofline_vocab, True, max(3, min(steps, 5)), 1, 77)
        b_s, b_row, b_nonpad = big[5], big[6], big[7]
        ach = n_pairs * (L * 8 + Q * 8 + (b_nonpad + Q) * b_row + 4) / b_s / 1e9
        hbm = {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "kernel_ms": b_s * 1e3, "mean_nonpad_terms_per_doc": b_nonpad,
               "leg": f"uniform term ids over a {args.roofline_vocab}-row table ({args.roofline_vocab * b_row / 1e9:.1f} GB of gathered rows), "
                      f"{n_queries} x {args.docs} pairs per launch: HBM is the binding resource"}
        del big
        _tables.pop((dev.index, args.roofline_vocab, D), None)
        torch.cuda.empty_cache()
    lists_view = None
    if as_lists and passes and world == 1:
        # the timed steps run the list route: its passes, each against the resource that binds it (lists_roofline); the pooling kernels of these
        # two models (per-lane top-k lists / MFMA convolutions + k-max) have no single-resource peak: their entry carries ms only
        docs = args.docs
        d2 = d_all.view(-1, docs * L)
        rows = float(sum(int((torch.unique(d2[i]) > 0).sum().item()) for i in range(d2.shape[0])))
        rstride = row // 4
        tokens = nonpad * n_pairs
        pool_name = {"drmmtks": "lists_tks_pool_kernel<12>", "pacrr": "pacrr_mfma_lists_kernel<5, 2>"}[model]
        ptab = [{"pass": "lists_clear_kernel (byte maps)", "ms": passes[0], "bytes_cleared": (n_pairs / docs) * ((V + 1023) // 1024 * 1024)},
                {"pass": "lists_mark_kernel", "ms": passes[1], "id_row_bytes": n_pairs * L * 8, "byte_stores": tokens},
                {"pass": "lists_query_kernel<5>", "ms": passes[2]},
                {"pass": "lists_sims_kernel<5, false>", "ms": passes[3], "rows_gathered": rows, "row_bytes": rows * rstride * 4, "fp32_fma": rows * Q * rstride, "pipe": SIMS_PIPE},
                {"pass": pool_name, "ms": passes[4]}]
        compulsory = n_pairs * (L * 8 + 4) + (n_pairs / docs) * Q * (8 + row) + int((torch.unique(d_all) > 0).sum().item()) * row
        lists_view = lists_roofline(model, {"passes": ptab, "kernel": headline_kernel}, None, n_pairs, kern_s, compulsory, None,
                                    "not measured for this leg (the KNRM / DRMM lines measure the shared passes)")
        lists_view.pop("headline_leg", None)
    rec = {
        "metric": "query-doc pairs scored/sec", "value": n_pairs * world * steps / elapsed, "unit": "pairs/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{rr.module_name} inference (SURVEY.md §8f row N4) on the KNRM benchmark's lists: qlen={Q} dlen={L} embed={D} vocab={V}, "
                               f"{args.docs} docs/query x {n_queries} queries per step per GPU, {'uniform' if args.uniform_ids else 'Zipf(1.1)'} term ids, "
                               + ("scored as whole candidate lists, " if as_lists else "")
                               + (f"consecutive steps round-robin over {n_streams} HIP streams, " if n_streams > 1 else "") + "reference default model options",
                   "pairs_per_step_per_gpu": n_pairs, "parallelism": f"query-sharded x{world}, one all_gather of scores per step" if world > 1 else "single GPU"},
        # `frac`: the HBM-bound leg (uniform ids over the --roofline-vocab table); on the Zipf ids of the headline leg the rate is a cache-level rate
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": hbm["achieved"] if hbm else (requested if args.uniform_ids else None), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": hbm["frac"] if hbm else (requested / HBM_PEAK_GBS if args.uniform_ids else None), "traffic": None,
                     "hbm_leg": hbm, "headline_kernel": headline_kernel,
                     "headline_route": "whole candidate lists (csrc/lists.hip); `kernel` / `frac` are the HBM-bound leg's per-pair kernel" if as_lists else "per-pair kernel",
                     "kernel_ms": kern_s * 1e3, "pairs_per_launch": n_pairs,
                     "requested_GBps": requested, "algorithmic_bytes_per_pair": abytes, "algorithmic_GBps": n_pairs * abytes / kern_s / 1e9,
                     "mean_nonpad_terms_per_doc": nonpad,
                     "note": "requested = int64 ids + one gathered row per in-vocabulary term / device time of one scoring call (one HIP event pair "
                             "around the timed steps); algorithmic = all L positions (pads are scored in closed form without a gather)"},
        "oracle_check": {"pairs": min(check_pairs, n_pairs), "max_err_of_scale": oracle_err},
    }
    if n_streams > 1:
        rec["step_streams"] = {"streams": n_streams}
    if lists_view is not None:      # the line's roofline = what its timed steps launch; the per-pair kernel's HBM-bound leg as the labelled secondary
        per_pair = rec["roofline"]
        per_pair["what"] = "SECONDARY, not what the timed steps launch: the one-pair-per-workgroup kernel on uniform ids over the --roofline-vocab table"
        lists_view["per_pair_hbm_leg"] = per_pair
        rec["roofline"] = lists_view
    if with_cpu and world == 1 and emb_h is not None:
        cores = os.cpu_count() or 1
        n = args.cpu_pairs or (2000 if model == "convknrm" else min(n_pairs, 2000 * max(1, cores // 4)))
        run = sibling_oracle(model, m, D, *(t[:n].cpu().numpy() for t in (q_all, d_all, idf_all)), emb_h)
        run()
        t0 = time.perf_counter()
        reps = 0
        while True:
            run()
            reps += 1
            if time.perf_counter() - t0 > 8.0 or reps >= 5:
                break
        rec["cpu_baseline"] = {"value": n * reps / (time.perf_counter() - t0), "unit": "pairs/s", "cores": cores, "kind": "port",
                               "sample": f"first {n} pairs of the step's batch, oracle/interaction_oracle.c with OpenMP over pairs ({reps} repetitions)"}
    return rec

