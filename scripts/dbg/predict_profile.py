"""cProfile of PytorchTrainer.predict's later calls (resident route) on the predict_e2e workload"""
import cProfile, pstats, os, sys
from types import SimpleNamespace
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from capreolus_amd import synthetic
from capreolus_amd.reranker import KNRM
from capreolus_amd.trainer.pytorch import PytorchTrainer
NQ, ND, V = 64, 1000, 400001
dev = torch.device("cuda:0")
emb = synthetic.make_embeddings(V, 300, seed=0)
cand = {k: v.cpu().numpy() for k, v in synthetic.make_candidate_list_torch(NQ, ND, V, dev).items()}
q2d = {str(q): [f"d{q}_{i}" for i in range(ND)] for q in range(NQ)}
row = {(str(q), f"d{q}_{i}"): q * ND + i for q in range(NQ) for i in range(ND)}
class PredData(torch.utils.data.IterableDataset):
    qid_to_docids = q2d
    def __iter__(self):
        for qid, docids in q2d.items():
            for d in docids:
                r = row[(qid, d)]
                yield {"qid": qid, "posdocid": d, "query": cand["query"][r], "posdoc": cand["posdoc"][r], "query_idf": cand["query_idf"][r]}
    def __len__(self): return NQ * ND
    def get_qid_docid_pairs(self):
        for qid, docids in q2d.items():
            for d in docids: yield qid, d
r = KNRM({}, SimpleNamespace(embeddings=emb)); r.build_model().to(dev).eval()
tr = PytorchTrainer({"evalbatch": 32, "lists": "always"}); tr.build()
s = PredData(); tr.predict(r, s); tr.predict(r, s)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): tr.predict(r, s)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
