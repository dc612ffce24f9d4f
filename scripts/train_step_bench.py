"""Training-step rate of the trainable rerankers (row N3): `reranker.score()` on a (query, positive, negative) batch of the reference's
default size (batch = 32, trainer/pytorch.py:24-27) -> pairwise hinge loss -> backward -> Adam step, on the GPU.  KNRM / DRMM / DRMM-TKS /
PACRR run their [B, Q, L] part on HIP feature kernels, ConvKNRM the reference's ATen op sequence under autograd.
Prints one JSON line per model; not part of bench.py's contract."""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capreolus_amd import synthetic  # noqa: E402
from capreolus_amd.reranker import DRMM, DRMMTKS, KNRM, PACRR, ConvKNRM  # noqa: E402
from capreolus_amd.trainer.pytorch import PytorchTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--vocab", type=int, default=400001)
ap.add_argument("--dim", type=int, default=300)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--only", default="", help="comma-separated model names")
args = ap.parse_args()
dev = torch.device("cuda:0")
emb = synthetic.make_embeddings(args.vocab, args.dim, seed=0)
cand = synthetic.make_candidate_list_torch(1, 2 * args.batch, args.vocab, dev)       # one query's candidates: the first half positives, the rest negatives
ext = SimpleNamespace(embeddings=emb, config={"maxqlen": 4}, pad=0)
B = args.batch
for name, r in (("KNRM", KNRM({}, ext)), ("DRMM", DRMM({}, ext)), ("DRMMTKS", DRMMTKS({}, ext)), ("PACRR", PACRR({}, ext)), ("ConvKNRM", ConvKNRM({}, ext))):
    if args.only and name not in args.only.split(","):
        continue
    fix = (lambda v: v.abs()) if name in ("ConvKNRM", "DRMM") else (lambda v: v)      # nn.Embedding ids only where the reference looks ids up unclamped
    d = {"query": fix(cand["query"][:B]), "query_idf": cand["query_idf"][:B], "posdoc": fix(cand["posdoc"][:B]), "negdoc": fix(cand["posdoc"][B:])}
    torch.manual_seed(0)
    m = r.build_model().to(dev).train()
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)

    def step():
        pos, neg = r.score(d)
        loss = torch.clamp(1.0 - (pos - neg), min=0).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.steps):
            last = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.steps, (time.perf_counter() - t0) * 1e3 / args.steps, last

    for _ in range(3):
        loss0 = step()
    eager_ms, eager_wall, loss = timed(step)
    rec = {"model": name, "batch": B, "docs_scored_per_step": 2 * B, "eager_ms_per_step": round(eager_ms, 3), "loss_first": round(float(loss0.detach()), 5),
           "loss_last_eager": round(float(loss.detach()), 5)}
    # (the eager steps' autograd graphs must be gone before a capture: their AccumulateGrad nodes belong to the default stream)
    del loss, loss0
    opt.zero_grad(set_to_none=True)
    del opt, m
    r = type(r)({}, ext)
    # the same step through the trainer's captured HIP graph (PytorchTrainer.single_train_iteration's route: one replay per batch)
    tr = PytorchTrainer({"batch": B, "itersize": B})
    tr.device, tr.scaler, tr.loss = dev, None, tr.pair_hinge_loss
    tr._train_graph, tr._graph_failed = None, False
    torch.manual_seed(0)
    m = r.build_model().to(dev).train()
    tr.optimizer = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=torch.tensor(1e-3, device=dev), capturable=True)
    first = tr._graphed_step(r, d)
    if first is None:
        rec["graph"] = "this model's step could not be captured"
    else:
        for _ in range(2):
            tr._graphed_step(r, d)
        g_ms, g_wall, gl = timed(lambda: tr._graphed_step(r, d))
        rec.update(ms_per_step=round(g_ms, 3), wall_ms_per_step=round(g_wall, 3), train_steps_per_s=round(1e3 / g_ms, 1), loss_last_graph=round(float(gl), 5))
    # ... and as the reranker's own fused step where it has one (capamd_{knrm,drmm,drmmtks}_train_step: two launches, plain Adam state updated in place)
    if callable(getattr(r, "fused_train_step", None)):
        del m
        r = type(r)({}, ext)
        torch.manual_seed(0)
        m = r.build_model().to(dev).train()
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
        from capreolus_amd import engine

        with engine.deferred_status(dev):
            first = r.fused_train_step(d, opt)
            if first is not None:
                for _ in range(2):
                    r.fused_train_step(d, opt)
                f_ms, f_wall, fl = timed(lambda: r.fused_train_step(d, opt))
                rec.update(fused_ms_per_step=round(f_ms, 3), fused_wall_ms_per_step=round(f_wall, 3), fused_train_steps_per_s=round(1e3 / max(f_ms, f_wall), 1),
                           loss_last_fused=round(float(fl), 5))
    print(json.dumps(rec))
