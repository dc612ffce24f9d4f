"""Random geometries through engine.NgramConv against the float64 op sequence (the body of tests/test_gpu_parity.py::test_ngram_conv_matches_conv1d)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from capreolus_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    N, Q, L = int(rng.integers(1, 10)), int(rng.integers(1, 9)), int(rng.integers(1, 260))
    D = int(rng.choice([4, 8, 52, 300, 316])); G = int(rng.integers(1, 5)); F = int(rng.choice([4, 8, 36, 128, 132, 256]))
    V = int(rng.integers(3, 400))
    emb = torch.tensor(rng.normal(0, 0.5, (V, D)).astype(np.float32), device=dev)
    q = rng.integers(0, V, (N, Q)); d = rng.integers(0, V, (N, L))
    for n in range(N):
        d[n, int(rng.integers(0, L + 1)):] = 0
    q, d = torch.tensor(q, device=dev), torch.tensor(d, device=dev)
    ws = [torch.tensor(rng.normal(0, 0.1, (F, D, g)).astype(np.float32), device=dev).requires_grad_() for g in range(1, G + 1)]
    bs = [torch.tensor(rng.normal(0, 0.1, (F,)).astype(np.float32), device=dev).requires_grad_() for g in range(1, G + 1)]
    wb = [t for p in zip(ws, bs) for t in p]
    qrep, drep = engine.NgramConv.apply(q, d, emb, *wb)
    gq = torch.tensor(rng.normal(0, 1, tuple(qrep.shape)).astype(np.float32), device=dev) * (q != 0)[:, None, :, None]
    gd = torch.tensor(rng.normal(0, 1, tuple(drep.shape)).astype(np.float32), device=dev) * (d != 0)[:, None, :, None]
    ((qrep * gq).sum() + (drep * gd).sum()).backward()
    w64 = [w.detach().double().requires_grad_() for w in ws]; b64 = [b.detach().double().requires_grad_() for b in bs]
    wq, wd = [], []
    for g in range(1, G + 1):
        for ids, out in ((q, wq), (d, wd)):
            out.append(Fn.conv1d(Fn.pad(emb.double()[ids].permute(0, 2, 1), (0, g - 1)), w64[g - 1], b64[g - 1]).permute(0, 2, 1))
    wq, wd = torch.stack(wq, 1), torch.stack(wd, 1)
    ((wq * gq.double()).sum() + (wd * gd.double()).sum()).backward()
    errs = []
    for have, want, ids in ((qrep, wq, q), (drep, wd, d)):
        real = (ids != 0)[:, None, :, None].expand_as(have)
        errs.append(float(((have.detach().double() - want.detach()).abs() * real).max()) / (float(want.detach().abs().max()) + 1e-30))
        if not bool(torch.isfinite(have).all()):
            errs.append(float("inf"))
    for have, want in zip([t.grad for t in wb], [t.grad for p in zip(w64, b64) for t in p]):
        errs.append(float((have.double() - want).abs().max()) / (float(want.abs().max()) + 1e-12))
    worst = max(errs)
    if not worst <= 5e-5:
        bad += 1
        print("MISMATCH", dict(N=N, Q=Q, L=L, D=D, G=G, F=F, V=V), worst)
print("ngram fuzz: %d mismatches" % bad)
