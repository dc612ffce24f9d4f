"""A/B of the S = 256 attention kernels on identical inputs (builder-side debugging aid): run once per CAPAMD_ATTN value, then compare the dumps.
  CAPAMD_ATTN=oneshot python scripts/dbg/attn_ab.py /tmp/a.pt 64;  python scripts/dbg/attn_ab.py /tmp/b.pt 64;  python scripts/dbg/attn_ab.py cmp /tmp/a.pt /tmp/b.pt"""
import sys

import torch

if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    S, H = 256, a.shape[1]
    d = (a.float() - b.float()).abs()
    print("equal:", torch.equal(a, b), "max abs diff", float(d.max()), "mismatching elements", int((d > 0).sum()), "of", d.numel())
    bad = (d > 0).nonzero()
    if len(bad):
        rows, cols = bad[:, 0], bad[:, 1]
        print("passages with mismatches:", torch.unique(rows // S).tolist()[:40])
        print("heads:", torch.unique(cols // 64).tolist(), " 32-row blocks inside a passage:", torch.unique((rows % S) // 32).tolist())
        print("d in head:", torch.unique(cols % 64).tolist()[:64])
    sys.exit(0)

from capreolus_amd import _lib   # noqa: E402

DEV = "cuda:0"
out, npsg = sys.argv[1], int(sys.argv[2])
S, hidden, heads = 256, 768, 12
g = torch.Generator(device=DEV).manual_seed(1)
M = npsg * S
x = torch.randn((M, hidden), generator=g, device=DEV).half()
w = (torch.randn((3 * hidden, hidden), generator=g, device=DEV) * 0.06).half()
b = torch.randn(3 * hidden, generator=g, device=DEV) * 0.1
lens = torch.randint(5, S + 1, (npsg,), generator=g, device=DEV)
mask = (torch.arange(S, device=DEV)[None, :] < lens[:, None]).long()
q, k, ctx = (torch.empty((M, hidden), dtype=torch.float16, device=DEV) for _ in range(3))
vt = torch.empty((npsg * heads, 64, S), dtype=torch.float16, device=DEV)
p = lambda t: t.data_ptr()
rc = _lib.load().capamd_bert_qkv_attention(p(x), p(w), p(b), p(mask), npsg, S, hidden, heads, p(q), p(k), p(vt), p(ctx), 1, torch.cuda.current_stream().cuda_stream)
assert rc == 0
torch.cuda.synchronize()
torch.save(ctx.cpu(), out)
