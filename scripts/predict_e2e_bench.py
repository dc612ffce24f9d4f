"""End to end through the trainer API (rows N1 / N2): one run of 64 queries x 1000 candidates (Q = 4, L = 800, D = 300) scored by KNRM
  (a) PytorchTrainer.predict            - the reference's route: PredSampler-style iterable -> DataLoader -> .to(device) -> test() per evalbatch,
                                          scores to the host, fp16 rounding and the {qid: {docid: score}} dict on the host
  (b) PytorchTrainer.predict_resident   - ids uploaded once into a CandidateStore, scored by index pairs (N1); same dict on the host
  (c) PytorchTrainer.evaluate_resident  - (b) + ranking and nDCG@20 on the device, one fp64 per query back (N2)
and checks that (a) and (b) return the same predictions.  Prints one JSON line; not part of bench.py's contract."""
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capreolus_amd import synthetic  # noqa: E402
from capreolus_amd.feeder import CandidateStore  # noqa: E402
from capreolus_amd.reranker import KNRM  # noqa: E402
from capreolus_amd.trainer.pytorch import PytorchTrainer  # noqa: E402

NQ, ND, V = 64, 1000, 400001
dev = torch.device("cuda:0")
emb = synthetic.make_embeddings(V, 300, seed=0)
cand = {k: v.cpu().numpy() for k, v in synthetic.make_candidate_list_torch(NQ, ND, V, dev).items()}
qid_to_docids = {str(q): [f"d{q}_{i}" for i in range(ND)] for q in range(NQ)}
row = {(str(q), f"d{q}_{i}"): q * ND + i for q in range(NQ) for i in range(ND)}


def id2vec(qid, docid):
    r = row[(qid, docid)]
    return {"qid": qid, "posdocid": docid, "query": cand["query"][r], "posdoc": cand["posdoc"][r], "query_idf": cand["query_idf"][r]}


class PredData(torch.utils.data.IterableDataset):   # PredSampler's contract (sampler/__init__.py:207-264)
    def __iter__(self):
        for qid, docids in qid_to_docids.items():
            for d in docids:
                yield id2vec(qid, d)

    def __len__(self):
        return NQ * ND

    def get_qid_docid_pairs(self):
        for qid, docids in qid_to_docids.items():
            for d in docids:
                yield qid, d

    qid_to_docids = qid_to_docids


rng = np.random.default_rng(0)
qrels = {q: {d: int(rng.integers(0, 3)) for d in rng.choice(ds, 20, replace=False)} for q, ds in qid_to_docids.items()}
r = KNRM({}, SimpleNamespace(embeddings=emb))
torch.manual_seed(0)
r.build_model().to(dev).eval()
out = {"workload": f"KNRM, {NQ} queries x {ND} candidates, Q=4 L=800 D=300"}


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        res = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps, res


for eb in (32, 1000):      # the reference's route, kept selectable: DataLoader -> .to(device) -> test() per (coalesced) batch
    tr = PytorchTrainer({"evalbatch": eb, "resident": False})
    tr.build()
    s, preds_a = timed(lambda: tr.predict(r, PredData()), reps=1)
    out[f"predict_dataloader_route_evalbatch{eb}_s"] = round(s, 3)
# the default route: the first predict() of a sampler walks it once and uploads its id rows, later calls score by index pairs
tr = PytorchTrainer({"evalbatch": 32})
tr.build()
sampler = PredData()
t0 = time.perf_counter()
preds_first = tr.predict(r, sampler)
torch.cuda.synchronize()
out["predict_first_call_s"] = round(time.perf_counter() - t0, 3)
s, preds_again = timed(lambda: tr.predict(r, sampler), reps=3)
out["predict_later_calls_s"] = round(s, 4)
out["predict_later_calls_pairs_per_s"] = round(NQ * ND / s)
# (the default route scores whole candidate lists - `lists` = "always" -: KNRM's pooling sums then run in another order than the DataLoader
# route's per-pair kernels, 1e-6 relative, so an fp16-rounded prediction can land on the other side of a rounding boundary)
dd = [(abs(preds_again[q][d] - preds_a[q][d]), abs(preds_a[q][d])) for q in preds_a for d in preds_a[q] if preds_again[q][d] != preds_a[q][d]]
out["predict_first_and_later_calls_identical"] = preds_first == preds_again
out["predict_default_vs_dataloader_route_differing_fp16_predictions"] = len(dd)
out["predict_default_vs_dataloader_route_max_rel_diff"] = max((x / max(y, 1e-6) for x, y in dd), default=0.0)
# `lists` = "always": KNRM as whole candidate lists too (its pooling sums in another order: 1e-6 relative, so an fp16-rounded prediction
# can land on the other side of a rounding boundary)
tr = PytorchTrainer({"evalbatch": 32, "lists": "never"})      # the per-pair kernels behind the same call
tr.build()
preds_l = tr.predict(r, sampler)
s, preds_l = timed(lambda: tr.predict(r, sampler), reps=3)
out["predict_per_pair_kernels_s"] = round(s, 4)
out["predict_per_pair_kernels_pairs_per_s"] = round(NQ * ND / s)
out["predict_per_pair_kernels_same_as_dataloader_route"] = preds_l == preds_a
t0 = time.perf_counter()
store = CandidateStore.from_id2vec(dev, qid_to_docids, id2vec)
out["store_upload_s"] = round(time.perf_counter() - t0, 3)
tr = PytorchTrainer({"evalbatch": 0})
tr.build()
s, preds_b = timed(lambda: tr.predict_resident(r, store, qid_to_docids))
out["predict_resident_s"] = round(s, 4)
s, ndcg = timed(lambda: tr.evaluate_resident(r, store, qid_to_docids, qrels, k=20))
out["evaluate_resident_s"] = round(s, 4)
out["ndcg_cut_20"] = round(ndcg, 6)
out["predict_resident_same_as_predict"] = preds_again == preds_b
out["pairs"] = NQ * ND
print(json.dumps(out))
