#!/usr/bin/env python
"""Regenerates tests/golden/*.npz by running the REFERENCE nn.Modules (imported from
/root/reference under the stub harness in _refharness.py) on seeded synthetic inputs.

Only runs in the build container (the reference tree does not travel to the GPU box); the
fixtures it writes are *data*: input ids, the tiny trainable weights, the seed of the embedding
table (regenerated bit-identically with numpy's legacy RandomState), and the reference's outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [knrm] [drmm] [bert]
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refharness  # noqa: E402
from capreolus_amd import synthetic  # noqa: E402


def _edge_cases(rs, batch, vocab):
    """Overwrite the first rows of a synthetic batch with the edge cases the contract names."""
    q, d = batch["query"], batch["posdoc"]
    B, Q = q.shape
    L = d.shape[1]
    # 0: full-length document, full-length query
    q[0] = synthetic.zipf_ids(rs, Q, vocab)
    d[0] = synthetic.zipf_ids(rs, L, vocab)
    # 1: all-pad document (negdoc placeholder, embedtext.py:151)
    d[1] = 0
    # 2: all-pad query
    q[2] = 0
    # 3: OOV exact match (negative ids equal) + unmatched OOV query term
    q[3, :] = 0
    q[3, 0] = -7
    q[3, 1] = 11
    if Q > 2:
        q[3, 2] = -9
    d[3, 5] = -7
    d[3, 6] = -7
    d[3, 9] = -8
    # 4: in-vocab exact matches (cos ~ 1.0), many of them
    q[4, 0] = 17
    d[4, 3:43:2] = 17
    # 5: single-term document
    d[5] = 0
    d[5, 0] = int(q[5][q[5] != 0][0]) if (q[5] != 0).any() else 3
    # 6: document made only of OOV terms
    d[6] = 0
    d[6, :10] = -np.arange(1, 11)
    # 7: pad in the middle of a doc (not produced by padlist, but the kernel must not assume it)
    d[7, 4] = 0
    batch["query_idf"] = np.where(q != 0, batch["query_idf"], 0).astype(np.float32)
    bad = (q != 0) & (batch["query_idf"] == 0)
    batch["query_idf"][bad] = 1.5
    return batch


def _multi_list_batch(rs, c):
    """several queries' candidate lists in one run (what `predict` scores): one list with an OOV query term some of its candidates contain,
    one with a single-term query"""
    n_lists, per = c["lists"], c["B"] // c["lists"]
    parts = [synthetic.make_candidate_list(rs, per, c["V"], c["Q"], c["L"], same_query=True, oov_range=40, query_oov_frac=0.0) for _ in range(n_lists)]
    for p in parts:                                     # idf is a property of the query: one row per list (the synthetic generator draws one per pair)
        p["query_idf"][:] = p["query_idf"][0]
    parts[2]["query"][:, 1] = -11                       # an OOV query term ...
    parts[2]["posdoc"][::7, 13] = -11                   # ... that every seventh candidate of that list contains (exact match, common.py:155-158)
    parts[5]["query"][:, 1:] = 0                        # a one-term query
    parts[5]["query_idf"][:, 1:] = 0
    return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}


def gen_knrm(KNRM, only=None):
    cases = {
        # name: (V, D, B, Q, L, config, seeds, perturb kernels?)
        "default": dict(V=5000, D=300, B=24, Q=4, L=800, cfg=dict(singlefc=True, scoretanh=False), pert=False),
        "twolayer_tanh": dict(V=3000, D=300, B=12, Q=4, L=800, cfg=dict(singlefc=False, scoretanh=True), pert=True),
        "glove50_short": dict(V=800, D=50, B=16, Q=3, L=100, cfg=dict(singlefc=True, scoretanh=True), pert=True),
        "dim100_q8": dict(V=1200, D=100, B=10, Q=8, L=230, cfg=dict(singlefc=False, scoretanh=False), pert=False),
        "ranklist": dict(V=20000, D=300, B=200, Q=4, L=800, cfg=dict(singlefc=True, scoretanh=False), pert=True, same_query=True),
        # what `predict` scores: several queries' candidate lists in one run (8 queries x 150 candidates; one query with an OOV term that
        # some of its candidates contain, one padded to a single term) - the fixture both scoring routes (per pair, whole lists) are
        # pinned on: the reference's fp16 scores and run order per query
        "multiquery": dict(V=20000, D=300, B=1200, Q=4, L=800, cfg=dict(singlefc=True, scoretanh=False), pert=True, same_query=True, lists=8),
    }
    for name, c in cases.items():
        if only and name not in only:
            continue
        seed = 100 + len(name)
        rs = np.random.RandomState(seed)
        emb = synthetic.make_embeddings(c["V"], c["D"], seed=seed)
        same = c.get("same_query", False)
        n_lists = c.get("lists", 1)
        if n_lists > 1:
            per = c["B"] // n_lists
            parts = [synthetic.make_candidate_list(rs, per, c["V"], c["Q"], c["L"], same_query=True, oov_range=40, query_oov_frac=0.0) for _ in range(n_lists)]
            parts[2]["query"][:, 1] = -11                       # an OOV query term ...
            parts[2]["posdoc"][::7, 13] = -11                   # ... that every seventh candidate of that list contains (exact match, common.py:155-158)
            parts[5]["query"][:, 1:] = 0                        # a one-term query
            parts[5]["query_idf"][:, 1:] = 0
            batch = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        else:
            batch = synthetic.make_candidate_list(rs, c["B"], c["V"], c["Q"], c["L"], same_query=same,
                                                  oov_range=40, query_oov_frac=0.0 if same else 0.1)
        if not same:
            batch = _edge_cases(rs, batch, c["V"])
        cfg = dict(gradkernels=True, finetune=False, **c["cfg"])
        torch.manual_seed(seed)
        model = KNRM.KNRM_class(SimpleNamespace(embeddings=emb), cfg).eval()
        if c["pert"]:  # "trained" kernels: mu/sigma are nn.Parameters (common.py:229-230)
            with torch.no_grad():
                for k in model.kernels.kernels:
                    k.mu.add_(float(rs.uniform(-0.02, 0.02)))
                    k.sigma.mul_(float(rs.uniform(0.9, 1.1)))
        q, d = torch.from_numpy(batch["query"]), torch.from_numpy(batch["posdoc"])
        with torch.no_grad():
            scores = model(d, q, torch.from_numpy(batch["query_idf"])).view(-1).numpy()
            sim = model.simmat(q, d).numpy()
        sd = {k: v.detach().numpy() for k, v in model.state_dict().items() if "embedding" not in k}
        out = dict(
            emb_seed=np.int64(seed), V=np.int64(c["V"]), D=np.int64(c["D"]),
            singlefc=np.bool_(cfg["singlefc"]), scoretanh=np.bool_(cfg["scoretanh"]),
            query=batch["query"].astype(np.int32), posdoc=batch["posdoc"].astype(np.int32),
            query_idf=batch["query_idf"], ref_scores=scores.astype(np.float32),
            ref_scores_f16=scores.astype(np.float16), ref_sim_rowsum=sim.sum(axis=2).astype(np.float32),
        )
        if n_lists > 1:
            out["list_offsets"] = np.arange(0, c["B"] + 1, c["B"] // n_lists, dtype=np.int64)
        for k, v in sd.items():
            out["sd." + k] = v
        np.savez_compressed(os.path.join(HERE, f"knrm_{name}.npz"), **out)
        print("knrm", name, scores[:6], "finite", np.isfinite(scores).all())


def gen_drmm(DRMM):
    cases = {
        "default": dict(V=5000, D=300, B=24, Q=4, L=800, cfg=dict(nbins=29, nodes=5, histType="LCH", gateType="IDF")),
        "zero_idf": dict(V=3000, D=300, B=12, Q=4, L=800, cfg=dict(nbins=29, nodes=5, histType="LCH", gateType="IDF"), zero_idf=True),
        "tv_nh": dict(V=800, D=50, B=16, Q=3, L=100, cfg=dict(nbins=11, nodes=7, histType="NH", gateType="TV")),
        "ch": dict(V=1200, D=100, B=10, Q=8, L=230, cfg=dict(nbins=29, nodes=5, histType="CH", gateType="IDF")),
        "ranklist": dict(V=20000, D=300, B=200, Q=4, L=800, cfg=dict(nbins=29, nodes=5, histType="LCH", gateType="IDF"), same_query=True),
    }
    for name, c in cases.items():
        seed = 200 + len(name)
        rs = np.random.RandomState(seed)
        emb = synthetic.make_embeddings(c["V"], c["D"], seed=seed)
        same = c.get("same_query", False)
        batch = synthetic.make_candidate_list(rs, c["B"], c["V"], c["Q"], c["L"], same_query=same, oov_range=40,
                                              query_oov_frac=0.0)
        if not same:
            batch = _edge_cases(rs, batch, c["V"])
            # DRMM cannot take OOV (negative) query ids: reference DRMM.py:109 indexes the embedding un-clamped
            batch["query"] = np.where(batch["query"] < 0, 0, batch["query"])
            batch["query_idf"] = np.where(batch["query"] != 0, batch["query_idf"], 0).astype(np.float32)
        if c.get("zero_idf"):
            batch["query_idf"][:] = 0  # default EmbedText behaviour (embedtext.py:86-87, 96)
        torch.manual_seed(seed)
        model = DRMM.DRMM_class(SimpleNamespace(embeddings=emb), dict(c["cfg"])).eval()
        with torch.no_grad():  # make the tiny weights non-degenerate ("trained")
            model.gates.weight.mul_(30.0)
            model.ffw[0].weight.mul_(4.0)
            model.ffw[2].weight.mul_(6.0)
        q, d = torch.from_numpy(batch["query"]), torch.from_numpy(batch["posdoc"])
        idf = torch.from_numpy(batch["query_idf"])
        with torch.no_grad():
            scores = model(d, q, idf).view(-1).numpy()
            sim = model.simmat(q, d).numpy()
            ht = model.hist_type
            model.hist_type = "CH"
            counts = model._hist_map(q, d, (d != 0).float()).numpy() - 1.0  # raw bin counts (DRMM.py:62-70)
            model.hist_type = ht
        # how many similarities sit within 4 ulp of a bin edge: these may flip a count
        edges = torch.linspace(-1, 1, c["cfg"]["nbins"] + 1)[1:].numpy()
        alle = np.concatenate([edges, np.float32([0.999, 1.001])])
        real = (batch["posdoc"] != 0)[:, None, :] & np.ones_like(sim, dtype=bool)
        ulp = np.spacing(np.abs(alle).astype(np.float32))
        near = (np.abs(sim[..., None] - alle) <= 4 * ulp) & real[..., None]
        n_amb = near.any(-1).sum(axis=(1, 2))
        sd = {k: v.detach().numpy() for k, v in model.state_dict().items() if "embedding" not in k}
        out = dict(
            emb_seed=np.int64(seed), V=np.int64(c["V"]), D=np.int64(c["D"]), nbins=np.int64(c["cfg"]["nbins"]),
            nodes=np.int64(c["cfg"]["nodes"]), histType=np.str_(c["cfg"]["histType"]), gateType=np.str_(c["cfg"]["gateType"]),
            query=batch["query"].astype(np.int32), posdoc=batch["posdoc"].astype(np.int32), query_idf=batch["query_idf"],
            ref_scores=scores.astype(np.float32), ref_scores_f16=scores.astype(np.float16),
            ref_counts=counts.astype(np.int32), n_ambiguous=n_amb.astype(np.int32), edges=edges.astype(np.float32),
        )
        for k, v in sd.items():
            out["sd." + k] = v
        np.savez_compressed(os.path.join(HERE, f"drmm_{name}.npz"), **out)
        print("drmm", name, scores[:6], "n_ambiguous", n_amb.tolist())


def gen_drmmtks(TKS, only=None):
    cases = {
        "default": dict(V=5000, D=300, B=24, Q=4, L=800, cfg=dict(topk=10, gateType="IDF", freezeemb=True)),
        "top3_short": dict(V=800, D=50, B=16, Q=3, L=100, cfg=dict(topk=3, gateType="IDF", freezeemb=True)),
        "ranklist": dict(V=20000, D=300, B=200, Q=4, L=800, cfg=dict(topk=10, gateType="IDF", freezeemb=True), same_query=True),
        # several queries' candidate lists in one run (8 x 100): what the whole-list route of `predict` is pinned on
        "multiquery": dict(V=20000, D=300, B=800, Q=4, L=800, cfg=dict(topk=10, gateType="IDF", freezeemb=True), same_query=True, lists=8),
    }
    for name, c in cases.items():
        if only and name not in only:
            continue
        seed = 300 + len(name)
        rs = np.random.RandomState(seed)
        emb = synthetic.make_embeddings(c["V"], c["D"], seed=seed)
        same = c.get("same_query", False)
        batch = _multi_list_batch(rs, c) if c.get("lists") else synthetic.make_candidate_list(rs, c["B"], c["V"], c["Q"], c["L"], same_query=same, oov_range=40,
                                              query_oov_frac=0.0 if same else 0.1)
        if not same:
            batch = _edge_cases(rs, batch, c["V"])
        torch.manual_seed(seed)
        model = TKS.DRMMTKS_class(SimpleNamespace(embeddings=emb), dict(c["cfg"])).eval()
        with torch.no_grad():
            model.gates.weight.mul_(30.0)
            model.ffw[0].weight.mul_(8.0)
        q, d = torch.from_numpy(batch["query"]), torch.from_numpy(batch["posdoc"])
        with torch.no_grad():
            scores = model(d, q, torch.from_numpy(batch["query_idf"])).view(-1).numpy()
        sd = {k: v.detach().numpy() for k, v in model.state_dict().items() if "embedding" not in k}
        out = dict(emb_seed=np.int64(seed), V=np.int64(c["V"]), D=np.int64(c["D"]), topk=np.int64(c["cfg"]["topk"]),
                   query=batch["query"].astype(np.int32), posdoc=batch["posdoc"].astype(np.int32), query_idf=batch["query_idf"],
                   ref_scores=scores.astype(np.float32), ref_scores_f16=scores.astype(np.float16))
        if c.get("lists"):
            out["list_offsets"] = np.arange(0, c["B"] + 1, c["B"] // c["lists"], dtype=np.int64)
        for k, v in sd.items():
            out["sd." + k] = v
        np.savez_compressed(os.path.join(HERE, f"drmmtks_{name}.npz"), **out)
        print("drmmtks", name, scores[:6], list(sd.keys()))


class _IdfTensor(torch.Tensor):
    """query_idf for the reference's PACRR forward: reshape(shape_tuple, 1) -> [*shape, 1] (see gen_pacrr)."""

    @staticmethod
    def __new__(cls, t):
        return torch.Tensor._make_subclass(cls, t)

    def reshape(self, *shape):
        flat = []
        for x in shape:
            flat.extend(list(x) if isinstance(x, (tuple, list, torch.Size)) else [x])
        return torch.Tensor.reshape(torch.Tensor._make_subclass(torch.Tensor, self), *flat)


def gen_pacrr(PACRR, only=None):
    cases = {
        "default": dict(V=5000, D=300, B=24, Q=4, L=800,
                        cfg=dict(mingram=1, maxgram=3, nfilters=32, idf=True, kmax=2, combine=32, nonlinearity="relu")),
        "tanh_noidf_short": dict(V=800, D=50, B=16, Q=3, L=100,
                                 cfg=dict(mingram=2, maxgram=3, nfilters=8, idf=False, kmax=3, combine=16, nonlinearity="tanh")),
        "ranklist": dict(V=20000, D=300, B=200, Q=4, L=800,
                         cfg=dict(mingram=1, maxgram=3, nfilters=32, idf=True, kmax=2, combine=32, nonlinearity="relu"), same_query=True),
        "multiquery": dict(V=20000, D=300, B=800, Q=4, L=800,
                           cfg=dict(mingram=1, maxgram=3, nfilters=32, idf=True, kmax=2, combine=32, nonlinearity="relu"), same_query=True, lists=8),
    }
    for name, c in cases.items():
        if only and name not in only:
            continue
        seed = 400 + len(name)
        rs = np.random.RandomState(seed)
        emb = synthetic.make_embeddings(c["V"], c["D"], seed=seed)
        same = c.get("same_query", False)
        batch = _multi_list_batch(rs, c) if c.get("lists") else synthetic.make_candidate_list(rs, c["B"], c["V"], c["Q"], c["L"], same_query=same, oov_range=40,
                                              query_oov_frac=0.0 if same else 0.1)
        if not same:
            batch = _edge_cases(rs, batch, c["V"])
        torch.manual_seed(seed)
        ext = SimpleNamespace(embeddings=emb, config={"maxqlen": c["Q"]})
        model = PACRR.PACRR_class(ext, dict(c["cfg"])).eval()
        with torch.no_grad():   # make every stage matter: larger conv weights, non-trivial biases
            for ng in model.ngrams:
                ng.conv.weight.mul_(3.0)
                ng.conv.bias.uniform_(-0.2, 0.3)
        q, d = torch.from_numpy(batch["query"]), torch.from_numpy(batch["posdoc"])
        with torch.no_grad():
            # PACRR.py:49 calls query_idf.reshape(query_idf.shape, 1), which raises TypeError under torch (shape tuple AND an int):
            # the reference's idf=True path (its default) cannot run as written.  The fixtures feed it a tensor subclass whose
            # reshape accepts that argument list and means what the line evidently intends - [B, Q] -> [B, Q, 1] - so the rest of
            # the reference's own code (softmax over dim 1, view, cat, combine) produces the expected scores.
            scores = model(d, q, _IdfTensor(torch.from_numpy(batch["query_idf"]))).view(-1).numpy()
        sd = {k: v.detach().numpy() for k, v in model.state_dict().items() if "embedding" not in k}
        out = dict(emb_seed=np.int64(seed), V=np.int64(c["V"]), D=np.int64(c["D"]), query=batch["query"].astype(np.int32),
                   posdoc=batch["posdoc"].astype(np.int32), query_idf=batch["query_idf"], ref_scores=scores.astype(np.float32),
                   ref_scores_f16=scores.astype(np.float16), nonlinearity=np.array(c["cfg"]["nonlinearity"]),
                   **{"cfg." + k: np.int64(v) for k, v in c["cfg"].items() if k != "nonlinearity"})
        if c.get("lists"):
            out["list_offsets"] = np.arange(0, c["B"] + 1, c["B"] // c["lists"], dtype=np.int64)
        for k, v in sd.items():
            out["sd." + k] = v
        np.savez_compressed(os.path.join(HERE, f"pacrr_{name}.npz"), **out)
        print("pacrr", name, scores[:6], list(sd.keys()))


def gen_convknrm(CONVKNRM):
    from tests.helpers import convknrm_conv_weights

    cases = {
        "default": dict(V=5000, D=300, B=12, Q=4, L=800,
                        cfg=dict(gradkernels=True, maxngram=3, crossmatch=True, filters=128, scoretanh=False, singlefc=True)),
        "nocross_2fc_short": dict(V=800, D=50, B=16, Q=3, L=60, perturb_kernels=True,
                                  cfg=dict(gradkernels=True, maxngram=2, crossmatch=False, filters=32, scoretanh=True, singlefc=False)),
        "ranklist": dict(V=20000, D=300, B=100, Q=4, L=800, same_query=True,
                         cfg=dict(gradkernels=True, maxngram=3, crossmatch=True, filters=128, scoretanh=False, singlefc=True)),
    }
    for name, c in cases.items():
        seed = 500 + len(name)
        rs = np.random.RandomState(seed)
        emb = synthetic.make_embeddings(c["V"], c["D"], seed=seed)
        same = c.get("same_query", False)
        batch = synthetic.make_candidate_list(rs, c["B"], c["V"], c["Q"], c["L"], same_query=same, oov_range=40, query_oov_frac=0.0)
        if not same:
            batch = _edge_cases(rs, batch, c["V"])
        # nn.Embedding takes ids in [0, V) only (the slowembedtext extractor has no negative OOV ids): fold the negatives in
        for k in ("query", "posdoc"):
            ids = batch[k].astype(np.int64)
            batch[k] = np.where(ids < 0, (-ids) % (c["V"] - 1) + 1, ids)
        torch.manual_seed(seed)
        ext = SimpleNamespace(embeddings=emb, pad=0)
        model = CONVKNRM.ConvKNRM_class(ext, dict(c["cfg"])).eval()
        ws, bs = convknrm_conv_weights(seed, c["cfg"]["filters"], c["D"], c["cfg"]["maxngram"])
        with torch.no_grad():
            for g, (w, b) in enumerate(zip(ws, bs)):
                model.convs[g][0].weight.copy_(torch.from_numpy(w))
                model.convs[g][0].bias.copy_(torch.from_numpy(b))
            model.combine[0].weight.mul_(4.0)
            if c.get("perturb_kernels"):
                for k in model.kernels.kernels:
                    k.mu.add_(0.03)
                    k.sigma.mul_(1.2)
            q, d = torch.from_numpy(batch["query"]), torch.from_numpy(batch["posdoc"])
            scores = model(d, q, torch.from_numpy(batch["query_idf"])).view(-1).numpy()
        sd = {k: v.detach().numpy() for k, v in model.state_dict().items() if "embeddings" not in k and not k.startswith("convs.")}
        out = dict(emb_seed=np.int64(seed), conv_seed=np.int64(seed), V=np.int64(c["V"]), D=np.int64(c["D"]), query=batch["query"].astype(np.int32),
                   posdoc=batch["posdoc"].astype(np.int32), query_idf=batch["query_idf"], ref_scores=scores.astype(np.float32),
                   ref_scores_f16=scores.astype(np.float16), **{"cfg." + k: np.int64(v) for k, v in c["cfg"].items()})
        for k, v in sd.items():
            out["sd." + k] = v
        np.savez_compressed(os.path.join(HERE, f"convknrm_{name}.npz"), **out)
        print("convknrm", name, scores[:6], list(sd.keys())[-4:])


if __name__ == "__main__":
    which = set(sys.argv[1:]) or {"knrm", "drmm", "bert", "drmmtks", "pacrr", "convknrm", "cedr"}
    common, KNRM, DRMM, MAXP, TKS, PACRR, CONVKNRM, CEDR = _refharness.load_reference()
    if "cedr" in which:
        from make_golden_bert import gen_cedr

        gen_cedr(CEDR)
    if "convknrm" in which:
        gen_convknrm(CONVKNRM)
    if "pacrr" in which or any(w.startswith("pacrr:") for w in which):
        gen_pacrr(PACRR, only={w[6:] for w in which if w.startswith("pacrr:")} or None)
    if "knrm" in which or any(w.startswith("knrm:") for w in which):     # "knrm:multiquery" = only that case
        gen_knrm(KNRM, only={w[5:] for w in which if w.startswith("knrm:")} or None)
    if "drmm" in which:
        gen_drmm(DRMM)
    if "drmmtks" in which or any(w.startswith("drmmtks:") for w in which):
        gen_drmmtks(TKS, only={w[8:] for w in which if w.startswith("drmmtks:")} or None)
    if "roberta" in which or (which >= {"knrm", "bert", "cedr"}):
        from make_golden_bert import gen_roberta

        gen_roberta(MAXP)
    if "bert" in which or any(w.startswith("bert:") for w in which):     # "bert:base_long" = only that case
        from make_golden_bert import gen_bert

        gen_bert(MAXP, only={w[5:] for w in which if w.startswith("bert:")} or None)
