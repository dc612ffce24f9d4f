#!/usr/bin/env python
"""Round-3 fixtures generated from the REFERENCE nn.Modules (same harness and rules as make_golden.py: build container only).

  drmm_<case>_alt.npz   the reference DRMM run AGAIN on the inputs of drmm_<case>.npz under different BLAS blockings - one thread,
                        one pair per call, pairs in reverse order, float64 `bmm` re-rounded - to show how far the reference
                        moves against ITSELF on the `sim < 1.0` coin flip of DRMM.py:62-66 (identical in-vocabulary terms have
                        cos in {1 - ulp, 1, 1 + ulp} depending on the summation order).
  <model>_grad_<case>.npz   .grad of every trainable parameter of the reference module after the reference trainer's pairwise hinge
                        loss (reranker/common.py:101-103) on (posdoc, negdoc) halves of the case's batch.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_extra.py [drmm_alt] [grad]
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


def _load(name):
    z = np.load(os.path.join(HERE, name + ".npz"))
    return {k: z[k] for k in z.files}


def _drmm_model(DRMM, fx):
    emb = synthetic.make_embeddings(int(fx["V"]), int(fx["D"]), seed=int(fx["emb_seed"]))
    cfg = dict(nbins=int(fx["nbins"]), nodes=int(fx["nodes"]), histType=str(fx["histType"]), gateType=str(fx["gateType"]))
    model = DRMM.DRMM_class(SimpleNamespace(embeddings=emb), cfg).eval()
    sd = model.state_dict()
    for k in list(sd):
        if "sd." + k in fx:
            sd[k] = torch.from_numpy(fx["sd." + k])
    model.load_state_dict(sd)
    return model


def _drmm_run(model, q, d, idf):
    with torch.no_grad():
        scores = model(d, q, idf).view(-1).numpy()
        ht = model.hist_type
        model.hist_type = "CH"
        counts = model._hist_map(q, d, (d != 0).float()).numpy() - 1.0
        model.hist_type = ht
    return scores.astype(np.float32), counts.astype(np.int32)


def gen_drmm_alt(DRMM):
    for name in ("default", "ranklist", "tv_nh", "ch"):
        fx = _load("drmm_" + name)
        model = _drmm_model(DRMM, fx)
        q, d = torch.from_numpy(fx["query"].astype(np.int64)), torch.from_numpy(fx["posdoc"].astype(np.int64))
        idf = torch.from_numpy(fx["query_idf"])
        base_s, base_c = _drmm_run(model, q, d, idf)
        same_as_fixture = bool(np.array_equal(base_c, fx["ref_counts"]) and np.array_equal(base_s, fx["ref_scores"]))
        out = {"regenerates_fixture": np.bool_(same_as_fixture)}
        n_thr = torch.get_num_threads()
        # (a) one thread
        torch.set_num_threads(1)
        s, c = _drmm_run(model, q, d, idf)
        torch.set_num_threads(n_thr)
        out["one_thread_scores"], out["one_thread_counts"] = s, c
        # (b) one pair per call (batch 1: another bmm blocking)
        ss, cc = zip(*[_drmm_run(model, q[i:i + 1], d[i:i + 1], idf[i:i + 1]) for i in range(q.shape[0])])
        out["batch1_scores"], out["batch1_counts"] = np.concatenate(ss), np.concatenate(cc)
        # (c) the batch reversed
        s, c = _drmm_run(model, q.flip(0), d.flip(0), idf.flip(0))
        out["reversed_scores"], out["reversed_counts"] = s[::-1].copy(), c[::-1].copy()
        # (d) round 4: more blockings - the batch in chunks of 2 / 4 / 8 / 32 pairs, 2 and 4 threads, denormals flushed
        for bs in (2, 4, 8, 32):
            ss, cc = zip(*[_drmm_run(model, q[i:i + bs], d[i:i + bs], idf[i:i + bs]) for i in range(0, q.shape[0], bs)])
            out[f"batch{bs}_scores"], out[f"batch{bs}_counts"] = np.concatenate(ss), np.concatenate(cc)
        for nt in (2, 4):
            torch.set_num_threads(nt)
            s, c = _drmm_run(model, q, d, idf)
            out[f"threads{nt}_scores"], out[f"threads{nt}_counts"] = s, c
        torch.set_num_threads(n_thr)
        flushed = torch.set_flush_denormal(True)
        s, c = _drmm_run(model, q, d, idf)
        torch.set_flush_denormal(False)
        if flushed:
            out["flush_denormal_scores"], out["flush_denormal_counts"] = s, c
        # (tried and left out: oneDNN disabled - ATen's other bmm path gives the same bits as the base run on all four cases)
        out["variants"] = np.array(sorted(k[:-7] for k in out if k.endswith("_scores")))
        for k in out["variants"]:
            moved = (out[k + "_counts"] != base_c).any(axis=(1, 2))
            rel = np.abs(out[k + "_scores"] - base_s) / np.maximum(np.abs(base_s), 1e-6)
            print(f"drmm_{name}_alt {k:10s}: pairs with moved counts {int(moved.sum())} of {len(moved)}, counts moved "
                  f"{int(np.abs(out[k + '_counts'] - base_c).sum() // 2)}, max rel score delta {rel.max():.3e}, pairs > 1e-3: {int((rel > 1e-3).sum())}")
        np.savez_compressed(os.path.join(HERE, f"drmm_{name}_alt.npz"), **out)


if __name__ == "__main__":
    which = set(sys.argv[1:]) or {"drmm_alt", "grad"}
    common, KNRM, DRMM, MAXP, TKS, PACRR, CONVKNRM, CEDR = _refharness.load_reference()
    if "drmm_alt" in which:
        gen_drmm_alt(DRMM)
    if "grad" in which:
        from make_golden_grad import gen_grads

        gen_grads(common, KNRM, DRMM, TKS, PACRR, CONVKNRM)
