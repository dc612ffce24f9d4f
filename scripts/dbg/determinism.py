"""Builder-side race hunt: the same call repeated many times must give bit-identical results every time (a kernel perturbing another
stream's timing runs alongside).  PYTHONPATH=. python scripts/dbg/determinism.py [attn|engine|smoke] [reps]"""
import sys
from types import SimpleNamespace

import numpy as np
import torch

from capreolus_amd import _lib, synthetic

DEV = "cuda:0"
what = sys.argv[1] if len(sys.argv) > 1 else "attn"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
noise_stream = torch.cuda.Stream(device=DEV)
na, nb = torch.randn((4096, 4096), device=DEV, dtype=torch.float16), torch.randn((4096, 4096), device=DEV, dtype=torch.float16)


def noise(i):
    if i % 3 == 0:
        return
    with torch.cuda.stream(noise_stream):
        for _ in range(i % 5):
            torch.mm(na, nb)


if what == "attn":
    S, hidden, heads = 256, 768, 12
    p = lambda t: t.data_ptr()
    for npsg in (3, 40, 43, 64, 129, 256):
        g = torch.Generator(device=DEV).manual_seed(npsg)
        M = npsg * S
        x = torch.randn((M, hidden), generator=g, device=DEV).half()
        w = (torch.randn((3 * hidden, hidden), generator=g, device=DEV) * 0.06).half()
        b = torch.randn(3 * hidden, generator=g, device=DEV) * 0.1
        lens = torch.randint(5, S + 1, (npsg,), generator=g, device=DEV)
        mask = (torch.arange(S, device=DEV)[None, :] < lens[:, None]).long()
        q, k, ctx = (torch.empty((M, hidden), dtype=torch.float16, device=DEV) for _ in range(3))
        vt = torch.empty((npsg * heads, 64, S), dtype=torch.float16, device=DEV)
        first, bad = None, 0
        for i in range(reps):
            noise(i)
            ctx.fill_(float("nan"))
            rc = _lib.load().capamd_bert_qkv_attention(p(x), p(w), p(b), p(mask), npsg, S, hidden, heads, p(q), p(k), p(vt), p(ctx), 1, torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            torch.cuda.synchronize()
            if first is None:
                first = ctx.clone()
            elif not torch.equal(first, ctx):
                bad += 1
        print(f"attention npsg {npsg}: {bad} of {reps - 1} repeats differ from the first", flush=True)
else:
    from capreolus_amd.reranker import PTBERTMaxP
    from oracle import bert_port

    if what == "smoke":
        dims = dict(hidden=128, layers=2, heads=2, ffn=512, vocab=1000, max_pos=128)
        B, P, S = 3, 3, 64
        wts = bert_port.random_weights(seed=11, **dims)
    else:
        dims = dict(hidden=768, layers=2, heads=12, ffn=3072, vocab=30522, max_pos=512)
        B, P, S = 160, 4, 256
        wts = synthetic.random_bert_weights(dims["hidden"], dims["layers"], dims["heads"], dims["ffn"], dims["vocab"], 512, seed=0)
    psg = synthetic.make_bert_passages(np.random.RandomState(11), B, P, S, vocab=dims["vocab"])
    for dt in ("fp16", "bf16"):
        r = PTBERTMaxP({"pretrained": dims, "compute_dtype": dt}, SimpleNamespace(config={"numpassages": P, "maxseqlen": S}))
        m = r.build_model()
        m.bert.load_state_dict(wts, strict=True)
        m.to(DEV).eval()
        d = {k: torch.as_tensor(v).to(DEV) for k, v in psg.items()}
        for skip in (True, False):
            first, bad = None, 0
            with torch.no_grad():
                for i in range(reps):
                    noise(i)
                    if i % 7 == 3:
                        m._engine._key = None   # re-pack the blob as a first call does
                    got = r.test(d) if skip else m._engine.forward(d["pos_bert_input"], d["pos_mask"], d["pos_seg"], "max", skip_padding=False)
                    torch.cuda.synchronize()
                    if first is None:
                        first = got.clone()
                    elif not torch.equal(first, got):
                        bad += 1
                        print("   rep", i, "max diff", float((first - got).abs().max()), flush=True)
            print(f"{what} {dt} skip_padding={skip}: {bad} of {reps - 1} repeats differ from the first", flush=True)
