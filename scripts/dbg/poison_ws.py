"""Builder-side check for reads of uninitialised workspace: run the BERT engine once on fresh workspaces, then overwrite every cached
workspace with 0xFF bytes (NaN in fp16 / bf16 / fp32) and run the same call again - the results must be bit-identical.
  PYTHONPATH=. python scripts/dbg/poison_ws.py [smoke|base] [fp16|bf16]"""
import sys
from types import SimpleNamespace

import numpy as np
import torch

from capreolus_amd import synthetic
from capreolus_amd.reranker import PTBERTMaxP
from oracle import bert_port

DEV = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "smoke"
dt = sys.argv[2] if len(sys.argv) > 2 else "fp16"
if which == "smoke":
    dims = dict(hidden=128, layers=2, heads=2, ffn=512, vocab=1000, max_pos=128)
    B, P, S = 3, 3, 64
else:
    dims = dict(hidden=768, layers=3, heads=12, ffn=3072, vocab=30522, max_pos=512)
    B, P, S = int(sys.argv[3]) if len(sys.argv) > 3 else 150, 4, 256
wts = bert_port.random_weights(seed=11, **dims) if which == "smoke" else synthetic.random_bert_weights(dims["hidden"], dims["layers"], dims["heads"], dims["ffn"], dims["vocab"], 512, seed=0)
psg = synthetic.make_bert_passages(np.random.RandomState(11), B, P, S, vocab=dims["vocab"])
r = PTBERTMaxP({"pretrained": dims, "compute_dtype": dt}, SimpleNamespace(config={"numpassages": P, "maxseqlen": S}))
m = r.build_model()
m.bert.load_state_dict(wts, strict=True)
m.to(DEV).eval()
d = {k: torch.as_tensor(v).to(DEV) for k, v in psg.items()}
for skip in (True, False):
    with torch.no_grad():
        r.test({k: v[:1] for k, v in d.items()})
        eng = m._engine
        clean = eng.forward(d["pos_bert_input"], d["pos_mask"], d["pos_seg"], "max", skip_padding=skip).clone()
        for rep in range(3):
            for k, ws in eng._wss.items():
                ws.fill_(0xFF)
            torch.cuda.synchronize()
            again = eng.forward(d["pos_bert_input"], d["pos_mask"], d["pos_seg"], "max", skip_padding=skip)
            ok = torch.equal(clean, again)
            print(f"{which} {dt} skip_padding={skip} rep {rep}: workspaces {list(eng._wss)} identical after poisoning: {ok}",
                  "" if ok else f"max diff {float((clean - again).abs().max())} nan {int(torch.isnan(again).sum())}")
