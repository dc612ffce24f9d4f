import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from capreolus_amd.engine import BertEngine
from oracle import bert_port
DEV = "cuda:0"
heads, layers, vocab = 4, 2, 500
for S, n_passages in [(96, 5), (160, 7), (224, 3), (64, 5)]:
    w = bert_port.random_weights(hidden=256, layers=layers, heads=heads, ffn=512, vocab=vocab, max_pos=S, seed=S + n_passages)
    g = torch.Generator().manual_seed(S * 3 + n_passages)
    n_full = 16
    ids = torch.randint(1, vocab, (n_full, 1, S), generator=g)
    lens = torch.randint(8, S + 1, (n_full, 1), generator=g)
    mask = (torch.arange(S)[None, None, :] < lens[:, :, None]).long()
    seg = ((torch.arange(S)[None, None, :] >= 6) & (mask > 0)).long()
    ids = ids * mask
    eng = BertEngine({k: v.to(DEV) for k, v in w.items()}, heads, compute_dtype="fp16", skip_padding=False)
    ref = bert_port.maxp(w, ids, mask, seg, heads, layers, "max").numpy()
    with torch.no_grad():
        full = eng.forward(ids.to(DEV), mask.to(DEV), seg.to(DEV), "max")
        torch.cuda.synchronize()
        print(S, "full16 err", np.abs(full.cpu().numpy() - ref).max(), flush=True)
        n = n_passages
        part = eng.forward(ids[:n].to(DEV), mask[:n].to(DEV), seg[:n].to(DEV), "max")
        torch.cuda.synchronize()
        print(S, "part err", np.abs(part.cpu().numpy() - ref[:n]).max(), "equal", torch.equal(part, full[:n]), (part - full[:n]).abs().max().item(), flush=True)
        eng.microbatch = 1
        split = eng.forward(ids[:n].to(DEV), mask[:n].to(DEV), seg[:n].to(DEV), "max")
        torch.cuda.synchronize()
        print(S, "split equal", torch.equal(split, full[:n]), (split - full[:n]).abs().max().item(), flush=True)
