"""Times ConvKNRM's convolution kernels (capreolus_amd/csrc/ngram_conv.hip) forward and backward at the training batch's shape:
N documents of L positions (N = 2 x batch), real lengths drawn like the bench's candidate lists or all positions real (--full)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capreolus_amd import engine, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=64)
ap.add_argument("--full", action="store_true")
ap.add_argument("--steps", type=int, default=50)
args = ap.parse_args()
dev = torch.device("cuda:0")
V, D, F, G = 400001, 300, 128, 3
emb = torch.randn(V, D, device=dev) * 0.3
cand = synthetic.make_candidate_list_torch(1, args.docs, V, dev)
q, d = cand["query"].abs(), cand["posdoc"].abs()
if args.full:
    d = torch.randint(1, V, d.shape, device=dev)
real = int((d != 0).sum())
ws = [(torch.randn(F, D, g, device=dev) * 0.05).requires_grad_() for g in range(1, G + 1)]
bs = [torch.zeros(F, device=dev).requires_grad_() for _ in range(G)]
wb = [t for p in zip(ws, bs) for t in p]
qrep, drep = engine.NgramConv.apply(q, d, emb, *wb)
gq, gd = torch.randn_like(qrep), torch.randn_like(drep) * (d != 0)[:, None, :, None]


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.steps * 1e3


fwd = timed(lambda: engine.NgramConv.apply(q, d, emb, *wb))
both = timed(lambda: torch.autograd.backward(engine.NgramConv.apply(q, d, emb, *wb), (gq, gd)))
flop = 2.0 * d.numel() * D * F * 6
print("docs %d x %d positions (%d real = %.0f %%): forward %.1f us, forward + backward %.1f us; all-position convolutions = %.1f GFLOP per direction -> forward %.1f TFLOP/s of them"
      % (d.shape[0], d.shape[1], real, 100.0 * real / d.numel(), fwd, both, flop / 1e9, flop / fwd / 1e6))
