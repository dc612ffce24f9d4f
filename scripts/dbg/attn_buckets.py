"""Attention kernels of every length bucket on 256 passages (builder-side aid; run under rocprofv3 --kernel-trace --stats and read the per-kernel averages)."""
import torch

from capreolus_amd import _lib

DEV = "cuda:0"
hidden, heads, npsg = 768, 12, 256
lib = _lib.load()
g = torch.Generator(device=DEV).manual_seed(1)
p = lambda t: t.data_ptr()
for S in range(32, 257, 32):
    M = npsg * S
    x = torch.randn((M, hidden), generator=g, device=DEV).half()
    w = (torch.randn((3 * hidden, hidden), generator=g, device=DEV) * 0.06).half()
    b = torch.randn(3 * hidden, generator=g, device=DEV) * 0.1
    mask = torch.ones((npsg, S), dtype=torch.long, device=DEV)
    q, k, ctx = (torch.empty((M, hidden), dtype=torch.float16, device=DEV) for _ in range(3))
    vt = torch.empty((npsg * heads, 64, S), dtype=torch.float16, device=DEV)
    for _ in range(5):
        rc = lib.capamd_bert_qkv_attention(p(x), p(w), p(b), p(mask), npsg, S, hidden, heads, p(q), p(k), p(vt), p(ctx), 1, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
