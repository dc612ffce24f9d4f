"""TREC run files and trec_eval-compatible nDCG@k: the two callers' formats either side of the
scoring path (reference capreolus/searcher/__init__.py:29-58; evaluator.py:55-85 delegates the
metric to pytrec_eval's ``ndcg_cut``, whose published definition is restated here)."""
import math
from collections import OrderedDict


def write_trec_run(preds, outfn, mode="wt"):
    """qids in integer order; per query by score descending, ties in insertion order
    (Python's stable sort, as the reference, searcher/__init__.py:48-58)."""
    with open(outfn, mode) as outf:
        for qid in sorted(preds.keys(), key=lambda k: int(k)):
            ranked = sorted(preds[qid].items(), key=lambda x: x[1], reverse=True)
            for rank, (docid, score) in enumerate(ranked, start=1):
                print(f"{qid} Q0 {docid} {rank} {score} capreolus", file=outf)


def load_trec_run(fn):
    run = OrderedDict()
    with open(fn, "rt") as f:
        for line in f:
            line = line.strip()
            if line:
                qid, _, docid, rank, score, desc = line.split()
                run.setdefault(qid, OrderedDict())[docid] = float(score)
    return run


def ndcg_cut(qrels, run, k=20):
    """trec_eval's ndcg_cut_k per query: gain = relevance level (negative -> 0), discount
    log2(rank+1), ideal ranking from the qrels; the run is ranked by score descending with ties broken
    by docid descending (trec_eval ignores the run's rank column).  Queries without qrels are skipped."""
    out = {}
    for qid, docs in run.items():
        rels = qrels.get(qid)
        if rels is None:
            continue
        ranked = sorted(docs.items(), key=lambda kv: (kv[1], kv[0]), reverse=True)[:k]
        dcg = sum(max(rels.get(d, 0), 0) / math.log2(i + 2) for i, (d, _) in enumerate(ranked))
        ideal = sorted((r for r in rels.values() if r > 0), reverse=True)[:k]
        idcg = sum(r / math.log2(i + 2) for i, r in enumerate(ideal))
        out[qid] = dcg / idcg if idcg > 0 else 0.0
    return out


def mean_ndcg_cut(qrels, run, k=20):
    v = ndcg_cut(qrels, run, k)
    return sum(v.values()) / len(v) if v else 0.0
