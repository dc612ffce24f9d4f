"""Ranking of scored candidate lists on the device.

What the reference does on the host after `reranker.test()` — `score.astype(np.float16)` (trainer/pytorch.py:346-348),
`Searcher.write_trec_run`'s stable per-query sort (searcher/__init__.py:48-58) and trec_eval's nDCG@k behind
`evaluator.eval_runs` (evaluator.py:55-85) — computed by `capamd_rank_candidates` / `capamd_ndcg_cut` on scores that
never leave HBM.  The host-side twins (`capreolus_amd.run_io`) are the parity oracle of these kernels.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from .engine import _need_gpu, _stream, status_word

MAX_CANDIDATES = 16384


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def offsets_of(counts, device):
    off = np.zeros(len(counts) + 1, dtype=np.int64)
    np.cumsum(np.asarray(counts, dtype=np.int64), out=off[1:])
    return torch.from_numpy(off).to(device)


def rank_candidates(scores, offsets, k, max_candidates=None):
    """scores fp32 [total] (device), offsets int64 [nq+1] (device, CSR).  Returns (idx int32 [nq, k], score fp16 [nq, k]):
    per query the positions of its candidates in run-file order (rounded score descending, ties in list order; -1 pads)
    and their fp16-rounded scores."""
    _need_gpu(scores, offsets)
    scores = scores.contiguous().float()
    nq = offsets.numel() - 1
    if max_candidates is None:
        max_candidates = int((offsets[1:] - offsets[:-1]).max().item()) if nq > 0 else 0
    idx = torch.empty((nq, k), dtype=torch.int32, device=scores.device)
    f16 = torch.empty((nq, k), dtype=torch.float16, device=scores.device)
    st = status_word(scores.device)
    _lib.check(_lib.load().capamd_rank_candidates(_p(scores), _p(offsets), nq, max_candidates, k, _p(idx), _p(f16), _p(st.t), _stream()),
               "capamd_rank_candidates")
    return idx, f16


def ndcg_cut(scores, offsets, rel, tie, idcg, k=20, max_candidates=None):
    """nDCG@k per query (fp64 [nq], device) from device-resident scores; `rel`, `tie`, `idcg` from `eval_arrays`."""
    _need_gpu(scores, offsets)
    scores = scores.contiguous().float()
    nq = offsets.numel() - 1
    if max_candidates is None:
        max_candidates = int((offsets[1:] - offsets[:-1]).max().item()) if nq > 0 else 0
    out = torch.empty(nq, dtype=torch.float64, device=scores.device)
    st = status_word(scores.device)
    _lib.check(_lib.load().capamd_ndcg_cut(_p(scores), _p(offsets), _p(rel), _p(tie), _p(idcg), nq, max_candidates, k, _p(out), _p(st.t),
                                           _stream()), "capamd_ndcg_cut")
    return out


def eval_arrays(qid_to_docids, qrels, k, device):
    """Host-side, once per candidate set: for every (qid, docid) in list order its relevance level, its rank in the
    docid-descending order trec_eval breaks score ties with, and per query the ideal DCG@k over ALL its judged documents.
    Queries without qrels get idcg 0 (their nDCG is reported as 0 and they should be left out of the mean, as
    run_io.ndcg_cut does)."""
    rel, tie, idcg, counts = [], [], [], []
    for qid, docids in qid_to_docids.items():
        judged = qrels.get(qid, {})
        order = sorted(range(len(docids)), key=lambda i: docids[i], reverse=True)
        ranks = [0] * len(docids)
        for r, i in enumerate(order):
            ranks[i] = r
        rel.extend(int(judged.get(d, 0)) for d in docids)
        tie.extend(ranks)
        ideal = sorted((r for r in judged.values() if r > 0), reverse=True)[:k]
        idcg.append(sum(r / math.log2(i + 2) for i, r in enumerate(ideal)))
        counts.append(len(docids))
    return (torch.tensor(rel, dtype=torch.int32, device=device), torch.tensor(tie, dtype=torch.int32, device=device),
            torch.tensor(idcg, dtype=torch.float64, device=device), offsets_of(counts, device))
