"""The BERT-base MaxP leg (BASELINE configs[3] / [4]): bf16 / fp16 MFMA encoder, MFMA roofline, its CPU baseline."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

from benchlib.common import MFMA_BF16_PEAK_TFLOPS, collective_info, timed_loop


def bert_flops_per_passage(S=256, H=768, F=3072, layers=12):
    """QKVO 4*2*S*H^2 + attention 2*2*S^2*H + FFN 2*2*S*H*F per layer."""
    return layers * (8 * S * H * H + 4 * S * S * H + 4 * S * H * F)


def bert_executed_flops_per_passage(S=256, H=768, F=3072, layers=12):
    """What the engine executes at full length: in the LAST layer only the [CLS] row of a passage is read afterwards, so its
    attention, output projection and FFN run on one row per passage (bert.hip); the QKV projection still covers all rows."""
    last_full = 2 * S * H * H + 4 * S * S * H + 4 * S * H * F                   # O-proj + attention + FFN of a whole layer
    last_cls = 2 * H * H + 4 * S * H + 4 * H * F                                # ... of one row
    return bert_flops_per_passage(S, H, F, layers) - last_full + last_cls


def bert_queries(n_docs_per_query, qids, P, S, VOCAB, dev):
    """BASELINE configs[3] / configs[4] generator: one query's candidate list per qid, seeded `1000 + qid`."""
    from capreolus_amd import synthetic

    parts = []
    for qid in qids:
        rs = np.random.RandomState(1000 + qid)
        host = synthetic.make_bert_passages(rs, min(n_docs_per_query, 64), P, S, vocab=VOCAB)
        reps = (n_docs_per_query + host["pos_bert_input"].shape[0] - 1) // host["pos_bert_input"].shape[0]
        d = {k: torch.as_tensor(np.tile(v, (reps, 1, 1))[:n_docs_per_query]).to(dev) for k, v in host.items()}
        # vary the tiled copies so that no two documents are identical
        d["pos_bert_input"] = torch.where((d["pos_mask"] == 1) & (d["pos_bert_input"] > 999),
                                          (d["pos_bert_input"] + torch.arange(n_docs_per_query, device=dev)[:, None, None] * 7) % (VOCAB - 1000) + 1000,
                                          d["pos_bert_input"])
        parts.append(d)
    return {k: torch.cat([p[k] for p in parts]) for k in parts[0]}


def bench_bert(args, ctx, steps, warmup, with_cpu):
    """BASELINE.json configs[3]: BERT-base MaxP, 4 passages x 256 tokens per document, 1000 docs/query.
    `--scaling strong --queries Q`: configs[4]'s shape - the step's Q queries (1000 candidates each, generator seeded 1000 + qid)
    are divided over the ranks in contiguous blocks, one all_gather of the document scores per step."""
    from types import SimpleNamespace

    from capreolus_amd import _lib, engine, synthetic
    from capreolus_amd.reranker import PTBERTMaxP

    world, rank, dev, use_dist, dist = ctx.world, ctx.rank, ctx.dev, ctx.use_dist, ctx.dist
    P, S, H, F, LAYERS, HEADS, VOCAB = 4, 256, 768, 3072, 12, 12, 30522
    nq = (args.queries or 1) if args.model == "bert" else 1
    strong = args.scaling == "strong" and args.model == "bert"
    if strong and nq % world:
        raise SystemExit("--scaling strong needs --queries divisible by the number of GPUs")
    per_rank_q = nq // world if strong else nq
    first_q = rank * per_rank_q
    d = bert_queries(args.docs, range(first_q, first_q + per_rank_q), P, S, VOCAB, dev)
    docs = args.docs * per_rank_q
    weights = synthetic.random_bert_weights(H, LAYERS, HEADS, F, VOCAB, 512, seed=0)
    rr = PTBERTMaxP({"pretrained": dict(hidden=H, layers=LAYERS, heads=HEADS, ffn=F, vocab=VOCAB, max_pos=512), "microbatch": args.bert_microbatch,
                     "compute_dtype": args.bert_dtype, "skip_padding": bool(args.bert_skip_padding)},
                    SimpleNamespace(config={"numpassages": P, "maxseqlen": S}))
    m = rr.build_model()
    m.bert.load_state_dict(weights, strict=True)
    m.to(dev).eval()
    with torch.no_grad():
        rr.test({k: v[:8] for k, v in d.items()})   # builds the 16-bit blob
    m._engine.n_streams = max(1, args.bert_streams)
    eng = m._engine
    gathered = torch.empty(docs * world, dtype=torch.float32, device=dev) if use_dist else None
    out = [None]

    def step(_):
        out[0] = eng.forward(d["pos_bert_input"], d["pos_mask"], d["pos_seg"], "max", check=False)
        if use_dist:
            dist.all_gather_into_tensor(gathered, out[0])

    for i in range(warmup):
        step(i)
    serial = eng.n_streams == 1
    elapsed, dev_s = timed_loop(ctx, step, 0, steps)
    engine.status_word(dev).raise_if_set()
    assert torch.isfinite(out[0]).all()
    scores = out[0].clone()

    # dominant kernel: the FFN1 GEMM (folded LayerNorm + bias + GELU epilogue), timed by HIP events around each of its launches
    # (capamd_debug_ffn1_timing, capreolus_amd/csrc/capamd_profiling.h - a hook of the -DCAPAMD_PROFILING build of the library only, so
    # the timed steps above ran the product library) in one more step of the same batch, run strictly serially (kernels of concurrent
    # streams would stretch each other's durations)
    tot_ms, launches, rows = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
    with _lib.profiling_build() as lib:
        eng.n_streams = 1
        step(0)                       # (sizes the single-stream workspace; module load of the second library)
        torch.cuda.synchronize(dev)
        lib.capamd_debug_ffn1_timing(1)
        step(0)
        torch.cuda.synchronize(dev)
        assert torch.equal(out[0], scores), "the serial and the multi-stream step disagree"
        eng.n_streams = max(1, args.bert_streams)
        _lib.check(lib.capamd_debug_ffn1_timing_read(ctypes.byref(tot_ms), ctypes.byref(launches), ctypes.byref(rows)), "ffn1 timing")
        lib.capamd_debug_ffn1_timing(0)
    serial = False
    out[0] = scores
    gemm_s = tot_ms.value * 1e-3 / max(1, launches.value)
    gemm_tf = 2.0 * rows.value * F * H / (tot_ms.value * 1e-3) / 1e12 if tot_ms.value > 0 else 0.0
    Mg = rows.value // max(1, launches.value)
    coll = collective_info(ctx, docs)       # (every rank takes part)
    if rank != 0:
        return None
    psg_per_s = docs * P * world * steps / elapsed
    step_tf = psg_per_s / world * bert_executed_flops_per_passage() / 1e12   # executed, not nominal, FLOPs
    rec = {
        "metric": "query-doc pairs scored/sec", "value": docs * world * steps / elapsed, "unit": "pairs/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.bert_dtype, "data": "synthetic",
        "config": {"workload": f"BERT-base MaxP inference (BASELINE.json configs[{4 if strong else 3}]): {P} passages x {S} tokens per doc, {docs} docs per step "
                               f"per GPU ({per_rank_q} quer{'y' if per_rank_q == 1 else 'ies'} x {args.docs} candidates, generator seeded 1000 + qid), seeded "
                               f"random-init weights, {args.bert_dtype} MFMA operands and activations, fp32 accumulate/LayerNorm statistics/softmax",
                   "passages_per_s": psg_per_s, "streams": eng.n_streams,
                   "parallelism": f"queries in contiguous blocks over {world} ranks, one all_gather of document scores per step" if world > 1 else "single GPU"},
        "roofline": {"bound": "mfma", "kernel": f"{'gemm_pingpong_kernel' if os.environ.get('CAPAMD_GEMM_RING') == '0' else 'gemm_ring_kernel'}<folded LayerNorm + bias + GELU> (FFN1: mean M={Mg} N={F} K={H}; {launches.value} launches "
                               + ("in the timed steps)" if serial else "in one strictly serial step after the timed ones)"),
                     "achieved": gemm_tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tf / MFMA_BF16_PEAK_TFLOPS,
                     "traffic": None, "kernel_ms": gemm_s * 1e3, "device_ms_per_step": dev_s * 1e3,
                     "whole_step_achieved": step_tf, "whole_step_frac": step_tf / MFMA_BF16_PEAK_TFLOPS,
                     "whole_step_frac_nominal": psg_per_s / world * bert_flops_per_passage() / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                     "algorithmic_flops_per_passage": bert_flops_per_passage(),
                     "executed_flops_per_passage": bert_executed_flops_per_passage(),
                     "note": "whole_step_* = executed FLOPs (last layer: [CLS] rows only after the QKV projection) / step time; *_nominal prices "
                             "every passage at SURVEY §8(d)'s 45.90 GFLOP"},
    }
    if coll is not None:
        rec["collective"] = coll
    if world == 1 and not args.no_bert_other_dtype and not args.bert_skip_padding:
        # the same step with the other 16-bit operand type (short: 3 steps), so that one line carries both
        import copy

        other = copy.copy(args)
        other.bert_dtype = "fp16" if args.bert_dtype == "bf16" else "bf16"
        other.no_bert_other_dtype = True
        del m, eng, rr
        torch.cuda.empty_cache()
        o = bench_bert(other, ctx, 3, 1, with_cpu=False)
        rec["other_operand_type"] = {"dtype": other.bert_dtype, "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": 3,
                                     "whole_step_frac": o["roofline"]["whole_step_frac"], "whole_step_frac_nominal": o["roofline"]["whole_step_frac_nominal"],
                                     "ffn1_frac": o["roofline"]["frac"]}
    if args.bert_skip_padding:
        # the nominal FLOP count (every passage at S tokens) no longer describes the executed work: no whole-step MFMA figure
        rec["config"]["padding"] = "passages encoded in length buckets of 32 tokens (identical scores; rows beyond a passage's last token are not computed)"
        rec["roofline"]["whole_step_achieved"] = rec["roofline"]["whole_step_frac"] = rec["roofline"]["whole_step_frac_nominal"] = None
    if with_cpu and world == 1:
        n = args.cpu_pairs or 1
        cores = os.cpu_count() or 1
        torch.set_num_threads(min(cores, 64))
        from oracle import bert_port   # the CPU leg only

        hd = {k: v[:n].cpu() for k, v in d.items()}
        t0 = time.perf_counter()
        want = bert_port.maxp(weights, hd["pos_bert_input"], hd["pos_mask"], hd["pos_seg"], HEADS, LAYERS, "max", chunk=16)
        dt = time.perf_counter() - t0
        # (a sanity bound on one document's MaxP score under wide random weights - the parity tests proper are tests/test_gpu_bert.py)
        err = float((out[0][:n].cpu() - want).abs().max() / max(1.0, float(want.abs().max())))
        assert err <= (5e-2 if args.bert_dtype == "bf16" else 1e-2), f"BERT: the timed scores differ from the fp32 port's by {err}"
        rec["parity"] = {"dtype": args.bert_dtype, "documents": n, "max_score_error_of_scale_vs_fp32_port": err,
                         "note": "the timed scores of the step's first document(s) against oracle/bert_port.py (fp32) under these wide random-init weights; "
                                 "the parity tests proper (reference fixtures, both dtypes) are tests/test_gpu_bert.py"}
        rec["cpu_baseline"] = {"value": n / dt, "unit": "pairs/s", "cores": min(cores, 64), "kind": "port",
                               "sample": f"{n} document(s) ({n * P} passages) through oracle/bert_port.py (fp32 ATen ops, {min(cores, 64)} threads); the timed "
                                         f"GPU scores of these documents agree with it to {err:.1e}"}
        torch.set_num_threads(cores)
    return rec

