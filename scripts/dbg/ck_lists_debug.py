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
for cfg in ({"maxngram": 1, "filters": 32}, {"maxngram": 2, "filters": 32, "crossmatch": False}, {"maxngram": 2, "filters": 32}, {"maxngram": 3, "filters": 128}):
    for L, ndoc in ((200, 30), (200, 30), (200, 8), (120, 30)):
        Q = 3
        q = np.tile(rs.randint(1, V, size=(1, Q)), (ndoc, 1)).astype(np.int64)
        d = rs.randint(1, V, size=(ndoc, L)).astype(np.int64)
        d[:, L - L // 4:] = 0
        r = ConvKNRM(cfg, SimpleNamespace(embeddings=emb, config={"maxqlen": Q}, pad=0))
        torch.manual_seed(5)
        r.build_model().to(DEV).eval()
        b = {"query": torch.as_tensor(q).to(DEV), "posdoc": torch.as_tensor(d).to(DEV), "query_idf": torch.zeros(ndoc, Q, device=DEV)}
        with torch.no_grad():
            p = r.test(b); l = r.test_lists(b, np.array([0, ndoc])); l2 = r.test_lists(b, np.array([0, ndoc])); p2 = r.test(b)
            halves = r.test_lists(b, np.array([0, ndoc // 2, ndoc]))
        idx = torch.nonzero(p != l).view(-1).tolist()
        print(cfg, "L", L, "docs", ndoc, "max diff", float((p - l).abs().max()), "n differing", int((p != l).sum()), "lists twice equal", bool(torch.equal(l, l2)),
              "pairs twice equal", bool(torch.equal(p, p2)), "halves == whole", bool(torch.equal(halves, l)), "where", idx[:8])
