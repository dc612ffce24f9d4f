import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import numpy as np, torch
from capreolus_amd import synthetic
from capreolus_amd.reranker import ConvKNRM
DEV = "cuda:0"
V, D = 3000, 300
emb = synthetic.make_embeddings(V, D, seed=11)
rs = np.random.RandomState(0)
cfg = {"maxngram": 1, "filters": 32}
Q = 3
ndoc = V - 1
q = np.tile(rs.randint(1, V, size=(1, Q)), (ndoc, 1)).astype(np.int64)
d = np.zeros((ndoc, 8), np.int64); d[:, 0] = np.arange(1, V)
r = ConvKNRM(cfg, SimpleNamespace(embeddings=emb, config={"maxqlen": Q}, pad=0))
torch.manual_seed(5); r.build_model().to(DEV).eval()
b = {"query": torch.as_tensor(q).to(DEV), "posdoc": torch.as_tensor(d).to(DEV), "query_idf": torch.zeros(ndoc, Q, device=DEV)}
with torch.no_grad():
    p = r.test(b); l = r.test_lists(b, np.array([0, ndoc]))
bad = torch.nonzero(p != l).view(-1).cpu().numpy()
print("single-token docs: differing", len(bad), "of", ndoc, "tokens", (bad + 1)[:40], "rel", float(((p - l).abs() / p.abs()).max()))
# the same tokens in documents of 16 tokens (a full tile), token of interest at each position
