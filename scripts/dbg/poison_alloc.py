"""Builder-side check for reads of uninitialised device memory anywhere in the BERT engine (blob, folded constants, outputs): with
POISON=1 the torch caching allocator is first filled with 0xFF bytes (NaN in every float type), so that every later torch.empty hands
out poisoned memory; the scores must equal those of a run on fresh memory (POISON=0, dumped to a file by a first invocation).
  PYTHONPATH=. POISON=0 python scripts/dbg/poison_alloc.py smoke fp16 /tmp/a.pt; POISON=1 python scripts/dbg/poison_alloc.py smoke fp16 /tmp/a.pt"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

from capreolus_amd import synthetic
from capreolus_amd.reranker import PTBERTMaxP
from oracle import bert_port

DEV = "cuda:0"
which, dt, path = sys.argv[1], sys.argv[2], sys.argv[3]
poison = os.environ.get("POISON", "0") == "1"
if poison:
    junk = [torch.full((1 << 30,), 0xFF, dtype=torch.uint8, device=DEV) for _ in range(12)]
    torch.cuda.synchronize()
    del junk
if which == "smoke":
    dims = dict(hidden=128, layers=2, heads=2, ffn=512, vocab=1000, max_pos=128)
    B, P, S = 3, 3, 64
    wts = bert_port.random_weights(seed=11, **dims)
else:
    dims = dict(hidden=768, layers=3, heads=12, ffn=3072, vocab=30522, max_pos=512)
    B, P, S = 150, 4, 256
    wts = synthetic.random_bert_weights(dims["hidden"], dims["layers"], dims["heads"], dims["ffn"], dims["vocab"], 512, seed=0)
psg = synthetic.make_bert_passages(np.random.RandomState(11), B, P, S, vocab=dims["vocab"])
r = PTBERTMaxP({"pretrained": dims, "compute_dtype": dt}, SimpleNamespace(config={"numpassages": P, "maxseqlen": S}))
m = r.build_model()
m.bert.load_state_dict(wts, strict=True)
m.to(DEV).eval()
d = {k: torch.as_tensor(v).to(DEV) for k, v in psg.items()}
res = {}
with torch.no_grad():
    r.test({k: v[:1] for k, v in d.items()})
    eng = m._engine
    for skip in (True, False):
        res[skip] = eng.forward(d["pos_bert_input"], d["pos_mask"], d["pos_seg"], "max", skip_padding=skip).cpu()
if not poison:
    torch.save(res, path)
    print("saved", {k: v[:3].tolist() for k, v in res.items()})
else:
    ref = torch.load(path)
    for k in res:
        ok = torch.equal(ref[k], res[k])
        print(f"{which} {dt} skip_padding={k}: poisoned allocator gives identical scores: {ok}", "" if ok else f"max diff {float((ref[k] - res[k]).abs().max())} nan {int(torch.isnan(res[k]).sum())}")
