import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
from capreolus_amd import _lib
from capreolus_amd.reranker import PTBERTMaxP
dev = "cuda:0"
H, LAYERS, HEADS, F, VOCAB = 768, int(os.environ.get("LAYERS", "3")), 12, 3072, 30522
rr = PTBERTMaxP({"pretrained": dict(hidden=H, layers=LAYERS, heads=HEADS, ffn=F, vocab=VOCAB, max_pos=512), "microbatch": 256, "compute_dtype": "bf16", "skip_padding": False},
                SimpleNamespace(config={"numpassages": 4, "maxseqlen": 256}))
torch.manual_seed(0)
m = rr.build_model().to(dev).eval()
g = torch.Generator(device=dev).manual_seed(5)
B, P, S = 128, 4, 256
ids = torch.randint(1000, VOCAB, (B, P, S), generator=g, device=dev)
lens = torch.randint(40, 250, (B, P, 1), generator=g, device=dev)
mask = (torch.arange(S, device=dev)[None, None, :] < lens).long()
ids = ids * mask
seg = (torch.arange(S, device=dev)[None, None, :] >= 8).long().expand(B, P, S).contiguous()
d = {"pos_bert_input": ids, "pos_mask": mask, "pos_seg": seg}
with torch.no_grad():
    a = rr.test(d).clone(); torch.cuda.synchronize()
    if os.environ.get("PROF"):
        with _lib.profiling_build():
            a = rr.test(d).clone(); torch.cuda.synchronize()
np.save(sys.argv[1], a.cpu().numpy())
