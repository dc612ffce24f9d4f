"""Random ConvKNRM configurations / batch geometries: the HIP training route (engine.NgramConv + engine.KernelPool) against the reference's
op sequence under ATen autograd (`_forward_train_aten`): scores and every gradient."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from capreolus_amd.reranker import ConvKNRM  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    B, Q, L = int(rng.integers(1, 9)), int(rng.integers(1, 7)), int(rng.choice([1, 3, 17, 160, 161, 400, 800, 900]))
    D, V = int(rng.choice([8, 52, 300])), int(rng.integers(5, 300))
    cfg = {"gradkernels": True, "maxngram": int(rng.integers(1, 4)), "crossmatch": bool(rng.integers(0, 2)), "filters": int(rng.choice([4, 32, 128])),
           "scoretanh": bool(rng.integers(0, 2)), "singlefc": bool(rng.integers(0, 2))}
    if (cfg["maxngram"] if cfg["crossmatch"] else 1) * Q > 24:
        continue
    emb = rng.normal(0, 0.5, (V, D)).astype(np.float32)
    emb[0] = 0
    q = rng.integers(1, V, (B, Q)); d = rng.integers(1, V, (B, L))
    for n in range(B):
        d[n, int(rng.integers(1, L + 1)):] = 0
        q[n, int(rng.integers(1, Q + 1)):] = 0
    if B > 2:
        d[1] = 0
    q, d = torch.tensor(q, device=dev), torch.tensor(d, device=dev)
    out = {}
    for route in ("hip", "aten"):
        torch.manual_seed(it)
        r = ConvKNRM(cfg, SimpleNamespace(embeddings=emb, config={"maxqlen": Q}, pad=0))
        m = r.build_model().to(dev).train()
        fwd = m._forward_train if route == "hip" else m._forward_train_aten
        pos = fwd(d, q).view(-1)
        neg = fwd(torch.roll(d, 1, 0), q).view(-1)
        loss = torch.clamp(1.0 - (pos - neg), min=0).mean() + 0.01 * pos.sum()
        loss.backward()
        out[route] = (pos.detach().cpu().numpy(), {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None})
    (ph, gh), (pa, ga) = out["hip"], out["aten"]
    worst = float(np.abs(ph - pa).max() / (np.abs(pa).max() + 1e-9))
    where = "scores"
    for k, want in ga.items():
        if k.startswith("kernels.kernels.10."):
            continue
        e = float(np.abs(gh[k] - want).max() / (np.abs(want).max() + 1e-7))
        if e > worst:
            worst, where = e, k
    if not worst <= 2e-3 or not np.isfinite(worst):
        bad += 1
        print("MISMATCH", dict(B=B, Q=Q, L=L, D=D, V=V, **cfg), where, worst)
print("convknrm train fuzz: %d mismatches" % bad)
