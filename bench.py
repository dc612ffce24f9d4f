#!/usr/bin/env python
"""Benchmark of the reranker scoring hot path on MI355X (BASELINE.json metric:
query-doc pairs scored/sec at 1/2/4/8 GPUs).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic candidate lists already
resident in HBM: by default BASELINE.json configs[1] — KNRM inference, qlen 4, dlen 800,
GloVe-shaped 400,001 x 300 fp32 table, 1000 docs/query, 64 queries per step per GPU
(SURVEY.md §8d "Config 2").  Multi-GPU: queries are sharded over ranks (weak scaling: every
rank scores its own 64 queries per step) and each step ends with one RCCL all-gather of the
score vectors.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (guides: MI355X_MICROARCH.md "HBM3E peak BW")


def algorithmic_bytes_per_pair(model, Q, L, D):
    """ids int64 + one fp32 embedding row per term + fp32 score (+ idf for DRMM)."""
    b = L * (8 + 4 * D) + Q * (8 + 4 * D) + 4
    return b + (4 * Q if model == "drmm" else 0)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="knrm", choices=["knrm", "drmm", "bert", "drmmtks", "pacrr", "convknrm"],
                    help="knrm (default, BASELINE.json's metric) | drmm | bert | the row-N4 siblings drmmtks, pacrr, convknrm")
    ap.add_argument("--queries", type=int, default=64, help="queries per step per GPU")
    ap.add_argument("--docs", type=int, default=1000, help="candidate documents per query")
    ap.add_argument("--launch-docs", type=int, default=0, help="pairs per kernel launch (0 = whole step in one launch)")
    ap.add_argument("--vocab", type=int, default=400001)
    ap.add_argument("--dim", type=int, default=300)
    ap.add_argument("--uniform-ids", action="store_true", help="uniform instead of Zipf(1.1) term ids (HBM-bound case)")
    ap.add_argument("--resident", action="store_true", help="score through the device-resident int32 candidate store (row N1)")
    ap.add_argument("--bert-dtype", default="fp16", choices=["bf16", "fp16"], help="16-bit operand type of the BERT encoder")
    ap.add_argument("--bert-skip-padding", action="store_true",
                    help="BERT: encode passages in length buckets (multiples of 32 tokens) - identical scores, padded rows not computed. Off by "
                         "default here: the headline line times the reference's full 4 x 256-token computation")
    ap.add_argument("--bert-two-streams", action="store_true",
                    help="BERT, full-length mode: the engine's default of running two halves of a large batch on two streams (+2.6 %). "
                         "Off here so that the dominant kernel's HIP-event duration in `roofline` is not inflated by a concurrent kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the CPU baseline sample (0 = auto)")
    return ap.parse_args()


def emit(rec):
    """The ONE JSON line, and the last line of stdout: RCCL prints a version banner through C stdio, which would otherwise
    be flushed at process exit, after Python's own line."""
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(json.dumps(rec), flush=True)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("CAPAMD_FORCE_DIST") == "1"  # the env knob exercises the RCCL path on one rank
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.model == "bert":
        return bench_bert(args, world, rank, dev, use_dist)
    if args.model in ("drmmtks", "pacrr", "convknrm"):
        return bench_sibling(args, world, rank, dev, use_dist)

    from types import SimpleNamespace

    from capreolus_amd import engine, synthetic
    from capreolus_amd.reranker import DRMM, KNRM

    Q, L, V, D = 4, 800, args.vocab, args.dim
    n_pairs = args.queries * args.docs
    # table: seeded the same on every rank (replicated); data: seeded per rank
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    emb = torch.randn((V, D), generator=g, device=dev) * 0.4
    emb[0] = 0
    batch = synthetic.make_candidate_list_torch(args.queries, args.docs, V, dev, seed=1 + rank, maxqlen=Q, maxdoclen=L,
                                                uniform_ids=args.uniform_ids)
    if args.model == "drmm":
        batch["query"] = batch["query"].clamp(min=0)

    torch.manual_seed(0)
    stub = SimpleNamespace(embeddings=np.zeros((2, D), dtype=np.float32))
    rr = (KNRM if args.model == "knrm" else DRMM)({}, stub)
    m = rr.build_model().to(dev).eval()
    m.embedding = torch.nn.Embedding.from_pretrained(emb, freeze=True)
    w = m.embedding.weight
    packed = m._packed.get(w)
    out = torch.empty(n_pairs, dtype=torch.float32, device=dev)
    gathered = torch.empty(n_pairs * world, dtype=torch.float32, device=dev) if use_dist else None

    launch = args.launch_docs or n_pairs
    slices = [(i, min(i + launch, n_pairs)) for i in range(0, n_pairs, launch)]
    q_all, d_all, idf_all = batch["query"], batch["posdoc"], batch["query_idf"]

    if args.model == "knrm":
        mu, sigma = m.kernels.stacked()
        w1, b1 = m.combine[0].weight.detach().contiguous(), m.combine[0].bias.detach()
        if args.resident:  # one query row per query, one document row per candidate, int32
            q_tab = q_all[:: args.docs].to(torch.int32).contiguous()
            d_tab = d_all.to(torch.int32).contiguous()
            pq = torch.arange(n_pairs, device=dev, dtype=torch.int32) // args.docs
            pd = torch.arange(n_pairs, device=dev, dtype=torch.int32)

            def launch_one(lo, hi):
                engine.knrm_forward_indexed(q_tab, d_tab, pq[lo:hi], pd[lo:hi], packed, V, D, mu, sigma, w1, b1, out=out[lo:hi], check=False)
        else:
            def launch_one(lo, hi):
                engine.knrm_forward(q_all[lo:hi], d_all[lo:hi], packed, V, D, mu, sigma, w1, b1, out=out[lo:hi], check=False)
    else:
        edges = m._bin_edges(dev)
        gw = m.gates.weight.detach().contiguous().view(-1)
        f0w, f0b = m.ffw[0].weight.detach().contiguous(), m.ffw[0].bias.detach()
        f2w, f2b = m.ffw[2].weight.detach().contiguous().view(-1), m.ffw[2].bias.detach()
        ow, ob = m.output_layer.weight.detach().view(-1), m.output_layer.bias.detach()

        def launch_one(lo, hi):
            engine.drmm_forward(q_all[lo:hi], d_all[lo:hi], idf_all[lo:hi], packed, V, D, edges, "LCH", "IDF", gw, w, f0w, f0b,
                                f2w, f2b, ow, ob, out=out[lo:hi], check=False)

    def step():
        for lo, hi in slices:
            launch_one(lo, hi)
        if use_dist:
            dist.all_gather_into_tensor(gathered, out)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    engine.status_word(dev).raise_if_set()
    assert torch.isfinite(out).all()
    if use_dist:
        assert torch.equal(gathered[rank * n_pairs:(rank + 1) * n_pairs], out)

    # ---- roofline of the dominant kernel: HIP events on the launch stream around each launch --
    evs = []
    for _ in range(max(3, min(args.steps, 10))):
        for lo, hi in slices:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch_one(lo, hi)
            e1.record()
            evs.append((e0, e1, hi - lo))
    torch.cuda.synchronize()
    kern_s = sum(e0.elapsed_time(e1) for e0, e1, _ in evs) * 1e-3 / len(evs)
    kern_pairs = sum(n for _, _, n in evs) / len(evs)
    abytes = algorithmic_bytes_per_pair(args.model, Q, L, D)
    achieved = kern_pairs * abytes / kern_s / 1e9
    nonpad = float((d_all > 0).sum().item()) / n_pairs
    real_bytes = (L * 8 + (nonpad + Q) * (4 * (packed.numel() // V)) + 4) * kern_pairs
    traffic = None
    tfile = os.path.join(ROOT, "profiles", f"{args.model}_hbm_traffic.json")
    if os.path.exists(tfile) and not args.uniform_ids and args.launch_docs == 0 and args.queries == 64:
        with open(tfile) as f:
            traffic = json.load(f).get("hbm_bytes_per_launch")

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    rec = {
        "metric": "query-doc pairs scored/sec",
        "value": n_pairs * world * args.steps / elapsed,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.model.upper()} inference (BASELINE.json configs[{1 if args.model == 'knrm' else 2}]): qlen={Q} dlen={L} "
                        f"embed={D} vocab={V}, {args.docs} docs/query x {args.queries} queries per step per GPU, "
                        f"{'uniform' if args.uniform_ids else 'Zipf(1.1)'} term ids, lognormal doc lengths, "
                        f"{len(slices)} launch(es) per step",
            "pairs_per_step_per_gpu": n_pairs,
            "parallelism": f"query-sharded x{world}, one all_gather of scores per step" if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": f"{args.model}_forward_kernel<5>",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_pair": abytes,
            "pairs_per_launch": kern_pairs,
            "kernel_ms": kern_s * 1e3,
            "note": "achieved = SURVEY §8(d) bytes (all L positions x fp32 row) / event-timed kernel duration; pads and OOV "
                    "terms are scored in closed form without a gather, so bytes actually requested are achieved_gathered",
            "achieved_gathered": real_bytes / kern_s / 1e9,
            "mean_nonpad_terms_per_doc": nonpad,
        },
    }

    if not args.no_cpu_baseline and world == 1:
        rec["cpu_baseline"] = cpu_baseline(args, m, batch, emb, Q, L, D)
    if use_dist:
        dist.destroy_process_group()
    emit(rec)


def bench_sibling(args, world, rank, dev, use_dist):
    """Row N4 models on the KNRM benchmark's candidate lists: DRMM-TKS, PACRR (KNRM's gather; same algorithmic bytes) and
    ConvKNRM (per position 6 projection-table parts of `filters` floats instead of one embedding row)."""
    from types import SimpleNamespace

    from capreolus_amd import engine, synthetic
    from capreolus_amd.reranker import DRMMTKS, PACRR, ConvKNRM

    if use_dist:
        import torch.distributed as dist
    Q, L, V, D = 4, 800, args.vocab, args.dim
    n_pairs = args.queries * args.docs
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    emb = torch.randn((V, D), generator=g, device=dev) * 0.4
    emb[0] = 0
    batch = synthetic.make_candidate_list_torch(args.queries, args.docs, V, dev, seed=1 + rank, maxqlen=Q, maxdoclen=L, uniform_ids=args.uniform_ids)
    if args.model == "convknrm":      # nn.Embedding ids only (the slowembedtext extractor has no negative OOV ids)
        batch = {k: (v.abs() if v.dtype == torch.int64 else v) for k, v in batch.items()}
    torch.manual_seed(0)
    stub = SimpleNamespace(embeddings=np.zeros((2, D), dtype=np.float32), config={"maxqlen": Q}, pad=0)
    rr = {"drmmtks": DRMMTKS, "pacrr": PACRR, "convknrm": ConvKNRM}[args.model]({}, stub)
    m = rr.build_model().to(dev).eval()
    name = "embeddings" if args.model == "convknrm" else "embedding"
    setattr(m, name, torch.nn.Embedding.from_pretrained(emb, freeze=True))
    q_all, d_all, idf_all = batch["query"], batch["posdoc"], batch["query_idf"]
    gathered = torch.empty(n_pairs * world, dtype=torch.float32, device=dev) if use_dist else None
    out = [None]

    def step():
        with torch.no_grad():
            out[0] = m(d_all, q_all, idf_all).view(-1)
        if use_dist:
            dist.all_gather_into_tensor(gathered, out[0])

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        m(d_all[:8], q_all[:8], idf_all[:8])          # packs the tables and checks the status word once, synchronously
    status = engine.deferred_status(dev)              # the timed calls are queued back to back like bench.py's KNRM / DRMM launches
    status.__enter__()                                # (check=False there); the accumulated status bits are raised at the end
    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # HIP events on the launch stream over the timed region
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    kern_s = ev0.elapsed_time(ev1) * 1e-3 / args.steps      # one scoring call = the model's kernel + a few tiny torch ops of the mirror
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out[0]).all()
    status.__exit__(None, None, None)
    nonpad = float((d_all != 0).sum().item()) / n_pairs
    if args.model == "convknrm":
        G, F = m.p["maxngram"], m.p["filters"]
        row = G * (G + 1) // 2 * F * 4
        abytes = L * (8 + row) + Q * (8 + row) + 4
        kname = "convknrm_forward_kernel<2>"
    else:
        row = 4 * D
        abytes = algorithmic_bytes_per_pair("knrm", Q, L, D) + 4 * Q
        kname = {"drmmtks": "drmmtks_forward_kernel<5>", "pacrr": "pacrr_mfma_kernel<5, 2>"}[args.model]
    achieved = n_pairs * abytes / kern_s / 1e9
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return
    rec = {
        "metric": "query-doc pairs scored/sec", "value": n_pairs * world * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{rr.module_name} inference (SURVEY.md §8f row N4) on the KNRM benchmark's lists: qlen={Q} dlen={L} embed={D} vocab={V}, "
                               f"{args.docs} docs/query x {args.queries} queries per step per GPU, {'uniform' if args.uniform_ids else 'Zipf(1.1)'} term ids, "
                               "reference default model options",
                   "pairs_per_step_per_gpu": n_pairs, "parallelism": f"query-sharded x{world}, one all_gather of scores per step" if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "algorithmic_bytes_per_pair": abytes, "pairs_per_launch": n_pairs, "kernel_ms": kern_s * 1e3,
                     "note": "achieved = (int64 ids + one gathered row per term, all L positions) / event-timed duration of one scoring call; pad "
                             "positions are scored in closed form without a gather, so the bytes actually requested are achieved_gathered",
                     "achieved_gathered": n_pairs * (L * 8 + (nonpad + Q) * row + 4) / kern_s / 1e9, "mean_nonpad_terms_per_doc": nonpad},
    }
    if not args.no_cpu_baseline and world == 1:
        from oracle import cpu as oracle   # the CPU leg only

        cores = os.cpu_count() or 1
        n = args.cpu_pairs or (2000 if args.model == "convknrm" else min(n_pairs, 2000 * max(1, cores // 4)))
        q, d, idf = (t[:n].cpu().numpy() for t in (q_all, d_all, idf_all))
        emb_h = emb.cpu().numpy()
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if "embedding" not in k}
        if args.model == "drmmtks":
            packed = oracle.pack(emb_h)

            def run():
                return oracle.drmmtks(q, d, idf, packed, D, m.topk, sd["gates.weight"], sd["ffw.0.weight"], sd["ffw.0.bias"],
                                      sd["output_layer.weight"], sd["output_layer.bias"])
        elif args.model == "pacrr":
            packed = oracle.pack(emb_h)
            p = m.p
            n_ng = p["maxgram"] - p["mingram"] + 1

            def run():
                return oracle.pacrr(q, d, idf, packed, D, p["mingram"], p["maxgram"], p["nfilters"], p["kmax"],
                                    [sd[f"ngrams.{i}.conv.weight"] for i in range(n_ng)], [sd[f"ngrams.{i}.conv.bias"] for i in range(n_ng)], p["idf"],
                                    sd["linear1.weight"], sd["linear1.bias"], sd["linear2.weight"], sd["linear2.bias"], sd["linear3.weight"],
                                    sd["linear3.bias"], p["nonlinearity"])
        else:
            p = m.p
            mu, sigma = (x.cpu().numpy() for x in m.kernels.stacked())

            def run():
                return oracle.convknrm(q, d, emb_h, [sd[f"convs.{i}.0.weight"] for i in range(p["maxngram"])],
                                       [sd[f"convs.{i}.0.bias"] for i in range(p["maxngram"])], p["crossmatch"], mu, sigma, sd["combine.0.weight"],
                                       sd["combine.0.bias"])
        want, err = run()
        assert err == 0
        got = out[0][:n].cpu().numpy()
        assert np.abs(got - want).max() <= 1e-3 * max(1.0, np.abs(want).max()), np.abs(got - want).max()   # the timed scores are the oracle's
        t0 = time.perf_counter()
        reps = 0
        while True:
            run()
            reps += 1
            if time.perf_counter() - t0 > 8.0 or reps >= 5:
                break
        rec["cpu_baseline"] = {"value": n * reps / (time.perf_counter() - t0), "unit": "pairs/s", "cores": cores, "kind": "port",
                               "sample": f"first {n} pairs of the step's batch, oracle/interaction_oracle.c with OpenMP over pairs ({reps} repetitions)"}
    if use_dist:
        dist.destroy_process_group()
    emit(rec)


MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 (MI355X_MICROARCH.md "Peak BF16/FP16 MFMA")


def bert_flops_per_passage(S=256, H=768, F=3072, layers=12):
    """QKVO 4*2*S*H^2 + attention 2*2*S^2*H + FFN 2*2*S*H*F per layer."""
    return layers * (8 * S * H * H + 4 * S * S * H + 4 * S * H * F)


def bert_executed_flops_per_passage(S=256, H=768, F=3072, layers=12):
    """What the engine executes at full length: in the LAST layer only the [CLS] row of a passage is read afterwards, so its
    attention, output projection and FFN run on one row per passage (bert.hip); the QKV projection still covers all rows."""
    last_full = 2 * S * H * H + 4 * S * S * H + 4 * S * H * F                   # O-proj + attention + FFN of a whole layer
    last_cls = 2 * H * H + 4 * S * H + 4 * H * F                                # ... of one row
    return bert_flops_per_passage(S, H, F, layers) - last_full + last_cls


def bench_bert(args, world, rank, dev, use_dist):
    """BASELINE.json configs[3]: BERT-base MaxP, 4 passages x 256 tokens per document, 1000 docs/query."""
    import ctypes
    from types import SimpleNamespace

    from capreolus_amd import _lib, engine, synthetic
    from capreolus_amd.reranker import PTBERTMaxP

    if use_dist:
        import torch.distributed as dist
    P, S, H, F, LAYERS, HEADS, VOCAB = 4, 256, 768, 3072, 12, 12, 30522
    docs = args.docs * (args.queries if args.queries != 64 else 1)   # default: one query's 1000 candidates per step
    rs = np.random.RandomState(1000 + rank)
    host = synthetic.make_bert_passages(rs, min(docs, 64), P, S, vocab=VOCAB)
    reps = (docs + host["pos_bert_input"].shape[0] - 1) // host["pos_bert_input"].shape[0]
    d = {k: torch.as_tensor(np.tile(v, (reps, 1, 1))[:docs]).to(dev) for k, v in host.items()}
    # vary the tiled copies so that no two documents are identical
    d["pos_bert_input"] = torch.where((d["pos_mask"] == 1) & (d["pos_bert_input"] > 999),
                                      (d["pos_bert_input"] + torch.arange(docs, device=dev)[:, None, None] * 7) % (VOCAB - 1000) + 1000,
                                      d["pos_bert_input"])
    weights = synthetic.random_bert_weights(H, LAYERS, HEADS, F, VOCAB, 512, seed=0)
    rr = PTBERTMaxP({"pretrained": dict(hidden=H, layers=LAYERS, heads=HEADS, ffn=F, vocab=VOCAB, max_pos=512), "microbatch": 256,
                     "compute_dtype": args.bert_dtype, "skip_padding": bool(args.bert_skip_padding)},
                    SimpleNamespace(config={"numpassages": P, "maxseqlen": S}))
    m = rr.build_model()
    m.bert.load_state_dict(weights, strict=True)
    m.to(dev).eval()
    with torch.no_grad():
        rr.test({k: v[:8] for k, v in d.items()})   # builds the bf16 blob
    m._engine.two_streams = bool(args.bert_two_streams)
    eng = m._engine
    gathered = torch.empty(docs * world, dtype=torch.float32, device=dev) if use_dist else None
    out = [None]

    def step():
        out[0] = eng.forward(d["pos_bert_input"], d["pos_mask"], d["pos_seg"], "max", check=False)
        if use_dist:
            dist.all_gather_into_tensor(gathered, out[0])

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    _lib.load().capamd_debug_ffn1_timing(1)  # HIP events around the dominant kernel's launches (read back below)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    engine.status_word(dev).raise_if_set()
    assert torch.isfinite(out[0]).all()

    # dominant kernel: the FFN1 GEMM (bias + GELU epilogue), timed by the library's HIP events around each of its launches
    # on the bench stream during the timed steps above (capamd_debug_ffn1_timing, include/capreolus_amd.h)
    lib = _lib.load()
    tot_ms, launches, rows = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.capamd_debug_ffn1_timing_read(ctypes.byref(tot_ms), ctypes.byref(launches), ctypes.byref(rows)), "ffn1 timing")
    lib.capamd_debug_ffn1_timing(0)
    gemm_s = tot_ms.value * 1e-3 / max(1, launches.value)
    gemm_tf = 2.0 * rows.value * F * H / (tot_ms.value * 1e-3) / 1e12 if tot_ms.value > 0 else 0.0
    Mg = rows.value // max(1, launches.value)
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return
    psg_per_s = docs * P * world * args.steps / elapsed
    step_tf = psg_per_s / world * bert_executed_flops_per_passage() / 1e12   # executed, not nominal, FLOPs
    rec = {
        "metric": "query-doc pairs scored/sec", "value": docs * world * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.bert_dtype, "data": "synthetic",
        "config": {"workload": f"BERT-base MaxP inference (BASELINE.json configs[3]): {P} passages x {S} tokens per doc, {docs} docs per step "
                               f"per GPU, seeded random-init weights, {args.bert_dtype} MFMA operands and activations, fp32 accumulate/LayerNorm statistics/softmax",
                   "passages_per_s": psg_per_s, "parallelism": f"document-sharded x{world}" if world > 1 else "single GPU"},
        "roofline": {"bound": "mfma", "kernel": f"gemm_pingpong_kernel<bias+GELU> (FFN1: mean M={Mg} N={F} K={H}; {launches.value} launches in the timed steps)",
                     "achieved": gemm_tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tf / MFMA_BF16_PEAK_TFLOPS,
                     "traffic": None, "kernel_ms": gemm_s * 1e3,
                     "whole_step_achieved": step_tf, "whole_step_frac": step_tf / MFMA_BF16_PEAK_TFLOPS,
                     "algorithmic_flops_per_passage": bert_flops_per_passage(),
                     "executed_flops_per_passage": bert_executed_flops_per_passage(),
                     "note": "whole_step_* = executed FLOPs (last layer: [CLS] rows only after the QKV projection) / step time"},
    }
    if args.bert_skip_padding:
        # the nominal FLOP count (every passage at S tokens) no longer describes the executed work: no whole-step MFMA figure
        rec["config"]["padding"] = "passages encoded in length buckets of 32 tokens (identical scores; rows beyond a passage's last token are not computed)"
        rec["roofline"]["whole_step_achieved"] = rec["roofline"]["whole_step_frac"] = None
    if not args.no_cpu_baseline and world == 1:
        n = args.cpu_pairs or 4
        cores = os.cpu_count() or 1
        torch.set_num_threads(cores)
        from oracle import bert_port   # the CPU leg only

        hd = {k: v[:n].cpu() for k, v in d.items()}
        t0 = time.perf_counter()
        bert_port.maxp(weights, hd["pos_bert_input"], hd["pos_mask"], hd["pos_seg"], HEADS, LAYERS, "max", chunk=16)
        dt = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": n / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
                               "sample": f"{n} documents ({n * P} passages) through oracle/bert_port.py (fp32 ATen ops, {cores} threads)"}
    if use_dist:
        dist.destroy_process_group()
    emit(rec)


def cpu_baseline(args, m, batch, emb, Q, L, D):
    """The CPU oracle (oracle/interaction_oracle.c, OpenMP over pairs) timed on this box's host cores on a
    bounded sample of the same workload; also the ATen op-sequence port for reference."""
    from oracle import cpu as oracle
    from oracle import torch_port

    cores = os.cpu_count() or 1
    n = args.cpu_pairs or min(batch["query"].shape[0], 2000 * max(1, cores // 4))
    q = batch["query"][:n].cpu().numpy()
    d = batch["posdoc"][:n].cpu().numpy()
    idf = batch["query_idf"][:n].cpu().numpy()
    emb_h = emb.cpu().numpy()
    packed = oracle.pack(emb_h)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if "embedding" not in k}
    if args.model == "knrm":
        mu, sigma = (x.cpu().numpy() for x in m.kernels.stacked())

        def run():
            return oracle.knrm(q, d, packed, D, mu, sigma, sd["combine.0.weight"], sd["combine.0.bias"])[0]
    else:
        edges = torch.linspace(-1, 1, 30)[1:].numpy()

        def run():
            return oracle.drmm(q, d, idf, packed, D, edges, "LCH", "IDF", sd["gates.weight"], emb_h, sd["ffw.0.weight"],
                               sd["ffw.0.bias"], sd["ffw.2.weight"], sd["ffw.2.bias"], sd["output_layer.weight"],
                               sd["output_layer.bias"])[0]

    run()  # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        run()
        reps += 1
        if time.perf_counter() - t0 > 8.0 or reps >= 5:
            break
    c_rate = n * reps / (time.perf_counter() - t0)

    # ATen port (what the reference executes on CPU): batches of 1000 pairs
    te = torch.as_tensor(emb_h)
    tq, td, tidf = torch.as_tensor(q), torch.as_tensor(d), torch.as_tensor(idf)
    torch.set_num_threads(cores)
    nb = min(n, 4000)
    with torch.no_grad():
        if args.model == "knrm":
            tmu, tsig = torch.as_tensor(mu), torch.as_tensor(sigma)
            tw, tb = torch.as_tensor(sd["combine.0.weight"]), torch.as_tensor(sd["combine.0.bias"])

            def trun(lo, hi):
                return torch_port.knrm(te, tq[lo:hi], td[lo:hi], tmu, tsig, tw, tb)
        else:
            ts = {k: torch.as_tensor(v) for k, v in sd.items()}

            def trun(lo, hi):
                return torch_port.drmm(te, tq[lo:hi], td[lo:hi], tidf[lo:hi], 29, "LCH", "IDF", ts["gates.weight"],
                                       ts["ffw.0.weight"], ts["ffw.0.bias"], ts["ffw.2.weight"], ts["ffw.2.bias"],
                                       ts["output_layer.weight"], ts["output_layer.bias"])
        trun(0, min(256, nb))
        t0 = time.perf_counter()
        done = 0
        while time.perf_counter() - t0 < 8.0:
            for lo in range(0, nb, 1000):
                trun(lo, min(lo + 1000, nb))
                done += min(lo + 1000, nb) - lo
                if time.perf_counter() - t0 > 8.0:
                    break
    t_rate = done / (time.perf_counter() - t0)
    # the same op sequence on ONE thread (SURVEY.md §8d asks for the single-thread figure too): 2 s
    torch.set_num_threads(1)
    with torch.no_grad():
        t0 = time.perf_counter()
        done1 = 0
        while time.perf_counter() - t0 < 2.0:
            trun(0, min(256, nb))
            done1 += min(256, nb)
    t1_rate = done1 / (time.perf_counter() - t0)
    torch.set_num_threads(cores)
    return {
        "value": c_rate,
        "unit": "pairs/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n} pairs of the step's batch, oracle/interaction_oracle.c with OpenMP over pairs ({reps} repetitions)",
        "aten_port_value": t_rate,
        "aten_port_note": f"oracle/torch_port.py (the reference's ATen op sequence) on {cores} threads, batches of 1000 pairs",
        "aten_port_single_thread_value": t1_rate,
    }


if __name__ == "__main__":
    main()
