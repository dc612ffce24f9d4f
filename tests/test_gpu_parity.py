"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference golden
vectors.  Needs an MI355X: `pytest -m gpu`."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from capreolus_amd import engine, run_io, synthetic
from capreolus_amd._lib import EngineError
from capreolus_amd.reranker import DRMM, KNRM
from oracle import cpu as oracle
from tests.helpers import (CONVKNRM_CASES, DRMM_CASES, KNRM_CASES, PACRR_CASES, REL_TOL, convknrm_args, convknrm_conv_weights, knrm_weights,
                           load_case, pacrr_args, rank_order, rel_err)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# GPU vs oracle on the continuous part (exp/log/tanh after bit-identical similarities): the two
# differ only by fp32 rounding of ~800-term sums and libm-vs-hardware transcendentals.
ORACLE_TOL = 2e-5


def _t(a):
    return torch.as_tensor(a).to(DEV)


def _knrm_model(c):
    cfg = {"singlefc": bool(c["singlefc"]), "scoretanh": bool(c["scoretanh"])}
    r = KNRM(cfg, SimpleNamespace(embeddings=c["emb"]))
    m = r.build_model()
    sd = {k[3:]: torch.as_tensor(v) for k, v in c.items() if k.startswith("sd.")}
    m.load_state_dict(sd, strict=False)  # the reference checkpoint format: everything but the embedding
    m.to(DEV).eval()
    return r


def _drmm_model(c):
    cfg = {"nbins": int(c["nbins"]), "nodes": int(c["nodes"]), "histType": str(c["histType"]), "gateType": str(c["gateType"])}
    r = DRMM(cfg, SimpleNamespace(embeddings=c["emb"]))
    m = r.build_model()
    sd = {k[3:]: torch.as_tensor(v) for k, v in c.items() if k.startswith("sd.")}
    m.load_state_dict(sd, strict=False)
    m.to(DEV).eval()
    return r


def _batch(c):
    return {"query": _t(c["query"]), "posdoc": _t(c["posdoc"]), "query_idf": _t(c["query_idf"])}


@pytest.mark.parametrize("V,D", [(257, 300), (100, 50), (64, 63), (33, 64), (50, 319), (40, 1), (77, 128)])
def test_pack_bit_exact(V, D):
    emb = synthetic.make_embeddings(V, D, seed=V + D)
    pe = engine.PackedEmbedding()
    got = pe.get(_t(emb)).cpu().numpy().reshape(V, -1)
    want = oracle.pack(emb)
    assert got.shape == want.shape
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


@pytest.mark.parametrize("name", KNRM_CASES)
def test_similarity_matrix_bit_exact(name):
    c = load_case("knrm", name)
    pe = engine.PackedEmbedding()
    packed = pe.get(_t(c["emb"]))
    got = engine.similarity_matrix(_t(c["query"]), _t(c["posdoc"]), packed, int(c["V"]), int(c["D"])).cpu().numpy()
    want, err = oracle.simmat(c["query"], c["posdoc"], oracle.pack(c["emb"]), int(c["D"]))
    assert err == 0
    assert (got.view(np.uint32) == want.view(np.uint32)).all(), np.abs(got - want).max()
    np.testing.assert_allclose(got.sum(axis=2, dtype=np.float64), c["ref_sim_rowsum"], rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("name", KNRM_CASES)
def test_knrm_scores(name):
    c = load_case("knrm", name)
    r = _knrm_model(c)
    with torch.no_grad():
        got = r.test(_batch(c)).cpu().numpy()
    assert got.shape == c["ref_scores"].shape
    mu, sigma, w1, b1, w2, b2 = knrm_weights(c)
    want, _ = oracle.knrm(c["query"], c["posdoc"], oracle.pack(c["emb"]), int(c["D"]), mu, sigma, w1, b1, w2, b2, bool(c["scoretanh"]))
    assert rel_err(got, want).max() <= ORACLE_TOL, rel_err(got, want).max()
    assert rel_err(got, c["ref_scores"]).max() <= REL_TOL, rel_err(got, c["ref_scores"]).max()


def test_knrm_rank_order():
    c = load_case("knrm", "ranklist")
    r = _knrm_model(c)
    with torch.no_grad():
        got = r.test(_batch(c)).cpu().numpy().astype(np.float16)  # trainer/pytorch.py:346-348
    # north star: "rank-order exactly".  All 200 fp16 scores equal the reference's bit for bit (observed 200 / 200: the closest any
    # reference score sits to an fp16 rounding boundary is 6.9e-6 relative, this kernel is within 1e-6 of it), hence the same run order.
    assert np.array_equal(got, c["ref_scores_f16"]), int((got != c["ref_scores_f16"]).sum())
    assert np.array_equal(rank_order(got), rank_order(c["ref_scores_f16"]))


@pytest.mark.parametrize("route", ["per_pair", "lists"])
def test_knrm_multiquery_run_matches_the_reference_by_either_route(route):
    """What `predict` scores - several queries' candidate lists in one run (8 x 150, one query with an OOV term some candidates contain,
    one of a single term) - against the REFERENCE's scores: both scoring routes of this engine, the per-pair kernels and the whole-list
    route (csrc/lists.hip, the trainer's default), reproduce the reference's fp16 predictions (`score.astype(np.float16)`,
    trainer/pytorch.py:346-348) and with them every query's run order.  A prediction may differ only where the reference's own fp32 score
    sits within 2e-6 (relative) of an fp16 rounding boundary - the closest one in this fixture is 5.2e-7 away - and then by one fp16 ulp."""
    c = load_case("knrm", "multiquery")
    r = _knrm_model(c)
    off = c["list_offsets"]
    with torch.no_grad():
        got = (r.test(_batch(c)) if route == "per_pair" else r.test_lists(_batch(c), off)).cpu().numpy()
    assert rel_err(got, c["ref_scores"]).max() <= 2e-5
    g16, r16 = got.astype(np.float16), c["ref_scores_f16"]
    bad = np.nonzero(g16 != r16)[0]
    ref = c["ref_scores"].astype(np.float64)
    for i in bad:        # only a reference score ON a rounding boundary (to 2e-6) may land on its other side
        mid = (g16[i].astype(np.float64) + r16[i].astype(np.float64)) / 2
        assert abs(ref[i] - mid) <= 2e-6 * abs(ref[i]) and abs(g16[i].view(np.int16).astype(int) - r16[i].view(np.int16).astype(int)) == 1, (i, got[i], ref[i])
    assert len(bad) <= 2, bad
    for a, b in zip(off[:-1], off[1:]):
        if not np.isin(bad, np.arange(a, b)).any():
            assert np.array_equal(rank_order(g16[a:b]), rank_order(r16[a:b]))


def _multiquery_sampler(c, qids=None):
    """the multi-query fixture behind the PredSampler contract (qid_to_docids, per-sample id rows); `qids`: a subset of its queries"""
    off = c["list_offsets"]
    q2d = {str(100 + k): [f"d{i}" for i in range(off[k], off[k + 1])] for k in range(len(off) - 1)}
    if qids is not None:
        q2d = {q: q2d[q] for q in qids}

    class Sampler(torch.utils.data.IterableDataset):
        qid_to_docids = q2d

        def __iter__(self):
            for qid, docs in self.qid_to_docids.items():
                for d in docs:
                    i = int(d[1:])
                    yield {"qid": qid, "posdocid": d, "query": c["query"][i].astype(np.int64), "posdoc": c["posdoc"][i].astype(np.int64), "query_idf": c["query_idf"][i]}

        def __len__(self):
            return sum(len(v) for v in self.qid_to_docids.values())

        def get_qid_docid_pairs(self):
            return ((q, d) for q, docs in self.qid_to_docids.items() for d in docs)

    return Sampler()


def test_knrm_predictions_do_not_depend_on_the_sharding():
    """`PytorchTrainer.predict` (defaults) on the reference's 8-query KNRM run: the fp16 predictions of every (query, document) are the
    SAME BITS whether the run is scored in one call, query by query (eight calls of one list), or as the shards `shard_pred_data` gives the
    ranks of a world of 2, 3 or 8 - the route is a function of the reranker and the configuration, not of how many lists a call holds
    (a rank holding one query used to take the per-pair kernels, whose pooling sums round differently).  The same with the
    DataLoader route (`resident` off: the per-pair kernels, whatever the batches' sizes) and with `lists` = "exact"."""
    from capreolus_amd.trainer import PytorchTrainer
    from capreolus_amd.trainer.pytorch import shard_pred_data

    c = load_case("knrm", "multiquery")
    r = _knrm_model(c)
    whole = _multiquery_sampler(c)
    qids = list(whole.qid_to_docids)
    for cfg in ({"batch": 32}, {"batch": 32, "resident": False}, {"batch": 32, "lists": "exact"}, {"batch": 32, "resident": False, "coalesce": 0}):
        base = PytorchTrainer(dict(cfg)).predict(r, whole)
        flat = np.array([base[q][d] for q in qids for d in whole.qid_to_docids[q]], dtype=np.float64)
        assert np.abs(flat - c["ref_scores"]).max() <= 2e-3 * np.abs(c["ref_scores"]).max()
        # query by query
        single = {}
        for q in qids:
            single.update(PytorchTrainer(dict(cfg)).predict(r, _multiquery_sampler(c, [q])))
        assert single == base, cfg
        # the shards of a world of 2 / 3 / 8 ranks, each scored by its own trainer
        for world in (2, 3, 8):
            merged = {}
            for rank in range(world):
                part, _, count, total = shard_pred_data(whole, rank, world)
                assert total == 1200
                if count:
                    merged.update(PytorchTrainer(dict(cfg)).predict(r, part))
            assert merged == base, (cfg, world)
    # ... and the routes agree with each other on this run's fp16 predictions except next to a rounding boundary
    a = PytorchTrainer({"batch": 32}).predict(r, whole)
    b = PytorchTrainer({"batch": 32, "resident": False}).predict(r, whole)
    diff = sum(a[q][d] != b[q][d] for q in qids for d in a[q])
    assert diff <= 2, diff


def test_knrm_predict_in_parts_equals_predict_by_query():
    """A run big enough for `predict` to score it in two / four overlapped parts (>= 16 lists, >= 16,000 pairs): the same fp16 predictions
    as scoring every query on its own."""
    from capreolus_amd.trainer import PytorchTrainer

    V, NQ, ND = 5000, 20, 1000
    rs = np.random.RandomState(3)
    emb = synthetic.make_embeddings(V, 300, seed=4)
    r = KNRM({}, SimpleNamespace(embeddings=emb))
    r.build_model().to(DEV).eval()
    lists = [synthetic.make_candidate_list(rs, ND, V, 4, 800, same_query=True, oov_range=30) for _ in range(NQ)]

    def sampler(which):
        q2d = {str(k): [f"q{k}d{i}" for i in range(ND)] for k in which}

        class Sampler(torch.utils.data.IterableDataset):
            qid_to_docids = q2d

            def __iter__(self):
                for qid, docs in self.qid_to_docids.items():
                    b = lists[int(qid)]
                    for i, d in enumerate(docs):
                        yield {"qid": qid, "posdocid": d, "query": b["query"][i], "posdoc": b["posdoc"][i], "query_idf": b["query_idf"][0]}

            def __len__(self):
                return ND * len(q2d)

            def get_qid_docid_pairs(self):
                return ((q, d) for q, docs in self.qid_to_docids.items() for d in docs)

        return Sampler()

    t = PytorchTrainer({"batch": 32})
    s = sampler(range(NQ))
    whole = t.predict(r, s)
    assert next(iter(t._resident_plans.values()))[2][4].get("parts"), "the run should have been scored in parts"
    again = t.predict(r, s)        # (the second call of a sampler: the kept plan)
    assert again == whole
    for k in (0, 7, NQ - 1):
        one = PytorchTrainer({"batch": 32}).predict(r, sampler([k]))
        assert one[str(k)] == whole[str(k)]


@pytest.mark.parametrize("kind", ["drmmtks", "pacrr"])
def test_multiquery_run_matches_the_reference_by_either_route(kind):
    """DRMM-TKS and PACRR on what `predict` scores - eight queries' candidate lists in one run (one query with an OOV term some candidates
    contain, one of a single term), fixtures from the REFERENCE modules: the whole-list route (the trainer's default) gives the per-pair
    kernels' scores bit for bit, both within 1e-3 of the reference's, and the fp16 predictions (`score.astype(np.float16)`,
    trainer/pytorch.py:346-348) - with them every query's run order - are the reference's except where its own fp32 score sits on an
    fp16 rounding boundary."""
    c = load_case(kind, "multiquery")
    r = _tks_reranker(c) if kind == "drmmtks" else _pacrr_reranker(c)
    off = c["list_offsets"]
    with torch.no_grad():
        pair = r.test(_batch(c)).cpu().numpy()
        got = r.test_lists(_batch(c), off).cpu().numpy()
    assert np.array_equal(pair, got)
    assert rel_err(got, c["ref_scores"]).max() <= REL_TOL
    g16, r16 = got.astype(np.float16), c["ref_scores_f16"]
    bad = np.nonzero(g16 != r16)[0]
    ref = c["ref_scores"].astype(np.float64)
    for i in bad:        # only a reference score next to a rounding boundary (to 1e-5 of itself: these models' fp32 sums differ from the reference's by that much) may land on its other side
        mid = (g16[i].astype(np.float64) + r16[i].astype(np.float64)) / 2
        assert abs(ref[i] - mid) <= 2e-5 * abs(ref[i]) + 1e-9 and abs(g16[i].view(np.int16).astype(int) - r16[i].view(np.int16).astype(int)) == 1, (i, got[i], ref[i])
    assert len(bad) <= 0.02 * len(got), len(bad)
    for a, b in zip(off[:-1], off[1:]):
        if not np.isin(bad, np.arange(a, b)).any():
            assert np.array_equal(rank_order(g16[a:b]), rank_order(r16[a:b]))


def test_drmm_lists_equal_the_per_pair_kernel_on_a_multi_query_run():
    """DRMM over eight queries' candidate lists with DIFFERENT queries and idf rows (the inputs of the DRMM-TKS multi-query fixture, its OOV
    query term replaced: DRMM.py:109 cannot take one): the whole-list route - a list scored against its first pair's query and idf row -
    gives the per-pair kernel's scores and matching histograms bit for bit."""
    from capreolus_amd.reranker import DRMM

    c = load_case("drmmtks", "multiquery")
    torch.manual_seed(21)
    r = DRMM({}, SimpleNamespace(embeddings=c["emb"]))
    r.build_model().to(DEV).eval()
    b = _batch(c)
    b["query"] = b["query"].clone()
    b["query"][b["query"] < 0] = 17
    off = c["list_offsets"]
    with torch.no_grad():
        pair = r.test(b)
        got = r.test_lists(b, off)
    assert torch.equal(pair, got) and torch.isfinite(got).all()
    assert float(got.std()) > 0       # (the lists' queries do differ: the scores are not one constant)


def test_lists_workspace_budget_only_changes_the_grouping(monkeypatch):
    """`engine.LISTS_WORKSPACE_BUDGET` bounds the per-list part of the whole-list workspace (17 B x V per list in flight): with room for a
    single list the library works through the lists one by one - same scores, bit for bit, as with all of them in flight."""
    c = load_case("knrm", "multiquery")
    r = _knrm_model(c)
    off = c["list_offsets"]
    with torch.no_grad():
        want = r.test_lists(_batch(c), off).clone()
        engine.release_workspaces()
        monkeypatch.setattr(engine, "LISTS_WORKSPACE_BUDGET", 17 * 20480 + 8192)       # one list's table + flags + query image
        got = r.test_lists(_batch(c), off)
        ws = next(iter(engine._list_workspaces.values()))
    assert torch.equal(got, want)
    assert ws.numel() < 1200 * (800 * 4 + 32) + 2 * (17 * 20480 + 8192)
    engine.release_workspaces()


def test_knrm_score_pair_interface():
    c = load_case("knrm", "default")
    r = _knrm_model(c)
    d = _batch(c)
    d["negdoc"] = torch.zeros_like(d["posdoc"])  # embedtext.py:151
    with torch.no_grad():
        pos, neg = r.score(d)
        ref = r.test(d)
    assert torch.equal(pos, ref)
    assert neg.shape == pos.shape and torch.isfinite(neg).all()


@pytest.mark.parametrize("name", DRMM_CASES)
def test_drmm_counts_and_scores(name):
    c = load_case("drmm", name)
    r = _drmm_model(c)
    B, Q = c["query"].shape
    counts = torch.empty((B, Q, int(c["nbins"]) + 1), dtype=torch.int32, device=DEV)
    b = _batch(c)
    with torch.no_grad():
        got = r.model(b["posdoc"], b["query"], b["query_idf"], counts_out=counts).view(-1).cpu().numpy()
        got2 = r.test(b).cpu().numpy()
    assert (got == got2).all()
    want, wcounts, err = oracle.drmm(
        c["query"], c["posdoc"], c["query_idf"], oracle.pack(c["emb"]), int(c["D"]), c["edges"], str(c["histType"]),
        str(c["gateType"]), c["sd.gates.weight"], c["emb"], c["sd.ffw.0.weight"], c["sd.ffw.0.bias"], c["sd.ffw.2.weight"],
        c["sd.ffw.2.bias"], c["sd.output_layer.weight"], c["sd.output_layer.bias"])
    assert err == 0
    assert (counts.cpu().numpy() == wcounts).all()  # integer work: bit exact
    assert rel_err(got, want).max() <= ORACLE_TOL, rel_err(got, want).max()
    # against the reference itself: exact counts / 1e-3 scores wherever no similarity is within 4 ulp
    # of a bin edge; elsewhere only the last regular bin may move (see tests/test_oracle_golden.py)
    d = counts.cpu().numpy().astype(np.int64) - c["ref_counts"].astype(np.int64)
    diff = np.abs(d).sum(axis=(1, 2))
    assert (diff[c["n_ambiguous"] == 0] == 0).all()
    assert (diff <= c["n_ambiguous"]).all()
    assert set(np.nonzero(np.abs(d).sum(axis=(0, 1)))[0].tolist()) <= {int(c["nbins"]) - 1}
    assert rel_err(got, c["ref_scores"])[diff == 0].max() <= REL_TOL


def test_errors_surface_as_exceptions():
    c = load_case("drmm", "default")
    r = _drmm_model(c)
    b = _batch(c)
    b["query"] = b["query"].clone()
    b["query"][0, 0] = -3
    with torch.no_grad(), pytest.raises(IndexError):
        r.test(b)
    c = load_case("knrm", "default")
    r = _knrm_model(c)
    b = _batch(c)
    b["posdoc"] = b["posdoc"].clone()
    b["posdoc"][1, 2] = int(c["V"])  # one past the table
    with torch.no_grad(), pytest.raises(IndexError):
        r.test(b)
    with torch.no_grad():  # and the engine is usable afterwards
        assert torch.isfinite(r.test(_batch(c))).all()
    with pytest.raises(RuntimeError):
        r.test({k: v.cpu() for k, v in _batch(c).items()})  # no CPU fallback
    r.model.train()
    with pytest.raises(RuntimeError):                # no CPU path in training either
        r.score({k: v.cpu() for k, v in {**_batch(c), "negdoc": _batch(c)["posdoc"]}.items()})


def test_empty_batch_and_weight_reload():
    c = load_case("knrm", "default")
    r = _knrm_model(c)
    b = _batch(c)
    with torch.no_grad():
        e = r.test({k: v[:0] for k, v in b.items()})
        assert e.shape == (0,)
        s0 = r.test(b).clone()
        r.model.combine[0].bias.add_(1.0)  # weights are read from the live module on every call
        s1 = r.test(b)
        assert torch.allclose(s1, s0 + 1.0, atol=1e-5)
        r.model.embedding.weight[5:].mul_(-1.0)  # in-place edit bumps the version -> re-pack
        s2 = r.test(b)
        assert torch.isfinite(s2).all()


# ---- BASELINE.json full sizes: size-independent properties + oracle on a sample -----------------
@pytest.fixture(scope="module")
def full():
    V, D = 400001, 300
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    emb = torch.randn((V, D), generator=g, device=DEV) * 0.4
    emb[0] = 0
    batch = synthetic.make_candidate_list_torch(2, 1000, V, DEV, seed=1)
    return emb, batch


def test_knrm_scores_do_not_depend_on_the_launch_size(full):
    """A pair's KNRM score is the same bits whichever per-pair kernel its launch's size selects: the streaming kernel (more than 3072
    pairs per launch), the one-pair-per-workgroup kernel with one row in flight per group (1537 .. 3072) or with four (up to 1536) -
    one shared tail (csrc/knrm.hip: knrm_pair_tail), the same folding of the sixteen groups' sums.  What `predict` stores is rounded to
    fp16 (reference trainer/pytorch.py:346-348): a score that moved by one fp32 ulp with the batch size could move a rank between a
    1-GPU and an 8-GPU run."""
    emb, _ = full
    batch = synthetic.make_candidate_list_torch(4, 1000, 400001, DEV, seed=11)
    torch.manual_seed(0)
    for cfg in ({}, {"singlefc": False}, {"scoretanh": True}):
        r = KNRM(cfg, SimpleNamespace(embeddings=np.zeros((2, 300), dtype=np.float32)))
        m = r.build_model().to(DEV).eval()
        m.embedding = torch.nn.Embedding.from_pretrained(emb, freeze=True)
        with torch.no_grad():
            whole = r.test(batch)                                    # 4000 pairs: the streaming kernel
            assert torch.isfinite(whole).all()
            for step in (2000, 1000, 1):
                n = 4000 if step > 1 else 7
                parts = torch.cat([r.test({k: v[i:i + step] for k, v in batch.items()}) for i in range(0, n, step)])
                assert torch.equal(parts, whole[:n]), (cfg, step, float((parts - whole[:n]).abs().max()))


def test_full_size_knrm_properties(full):
    emb, batch = full
    torch.manual_seed(0)      # (the combine layer's initial weights set the score scale the tolerances below are relative to)
    r = KNRM({}, SimpleNamespace(embeddings=np.zeros((2, 300), dtype=np.float32)))
    m = r.build_model().to(DEV).eval()
    m.embedding = torch.nn.Embedding.from_pretrained(emb, freeze=True)
    with torch.no_grad():
        s = r.test(batch)
        assert s.shape == (2000,) and torch.isfinite(s).all()
        # (a) pairs are independent: any permutation of the batch permutes the scores bit for bit
        perm = torch.randperm(2000, device=DEV)
        sp = r.test({k: v[perm] for k, v in batch.items()})
        assert torch.equal(sp, s[perm])
        # (b) chunked calls == one call (evalbatch does not matter)
        sc = torch.cat([r.test({k: v[i:i + 333] for k, v in batch.items()}) for i in range(0, 2000, 333)])
        assert torch.equal(sc, s)
        # (c) order of terms inside a document only changes fp32 summation order
        doc = batch["posdoc"].clone()
        L = doc.shape[1]
        doc = doc[:, torch.randperm(L, device=DEV)]
        s3 = r.test({**batch, "posdoc": doc})
        assert (s3 - s).abs().max() <= 2e-5 * s.abs().max()
    # (d) a sample of pairs against the oracle
    idx = np.arange(0, 2000, 63)
    q, d = batch["query"][idx].cpu().numpy(), batch["posdoc"][idx].cpu().numpy()
    used = np.unique(np.concatenate([q.ravel(), d.ravel()]))
    used = used[used > 0]
    remap = np.zeros(400001, dtype=np.int64)
    remap[used] = np.arange(1, len(used) + 1)
    small = np.concatenate([np.zeros((1, 300), np.float32), emb[torch.as_tensor(used, device=DEV)].cpu().numpy()])
    rq, rd = np.where(q > 0, remap[np.maximum(q, 0)], q), np.where(d > 0, remap[np.maximum(d, 0)], d)
    mu, sigma = (x.cpu().numpy() for x in m.kernels.stacked())
    want, _ = oracle.knrm(rq, rd, oracle.pack(small), 300, mu, sigma, m.combine[0].weight.detach().cpu().numpy(),
                          m.combine[0].bias.detach().cpu().numpy())
    assert rel_err(s[idx].cpu().numpy(), want).max() <= ORACLE_TOL


def test_full_size_drmm_properties(full):
    emb, batch = full
    torch.manual_seed(0)
    r = DRMM({}, SimpleNamespace(embeddings=np.zeros((2, 300), dtype=np.float32)))
    m = r.build_model().to(DEV).eval()
    m.embedding = torch.nn.Embedding.from_pretrained(emb, freeze=True)
    b = {k: v for k, v in batch.items()}
    b["query"] = b["query"].clamp(min=0)
    with torch.no_grad():
        c0 = torch.empty((2000, 4, 30), dtype=torch.int32, device=DEV)
        s = m(b["posdoc"], b["query"], b["query_idf"], counts_out=c0).view(-1)
        assert torch.isfinite(s).all()
        # every non-pad term falls in at most one regular bin
        nonpad = (b["posdoc"] != 0).sum(1)
        assert (c0[:, :, :29].sum(-1) <= nonpad[:, None]).all()
        # (a) term order inside the document: integer counts are order independent -> bit-identical scores
        doc = b["posdoc"][:, torch.randperm(800, device=DEV)]
        c1 = torch.empty_like(c0)
        s1 = m(doc, b["query"], b["query_idf"], counts_out=c1).view(-1)
        assert torch.equal(c0, c1) and torch.equal(s, s1)
        # (b) extra pad positions change nothing (DRMM.py:57 pushes pads out of every bin)
        wide = torch.cat([b["posdoc"], torch.zeros((2000, 133), dtype=torch.int64, device=DEV)], 1)
        s2 = m(wide, b["query"], b["query_idf"]).view(-1)
        assert torch.equal(s, s2)
    # (c) a stride-63 sample of pairs against the oracle: bin counts bit-exact, scores to ORACLE_TOL
    idx = np.arange(0, 2000, 63)
    q, d = b["query"][idx].cpu().numpy(), b["posdoc"][idx].cpu().numpy()
    used = np.unique(np.concatenate([q.ravel(), d.ravel()]))
    used = used[used > 0]
    remap = np.zeros(400001, dtype=np.int64)
    remap[used] = np.arange(1, len(used) + 1)
    small = np.concatenate([np.zeros((1, 300), np.float32), emb[torch.as_tensor(used, device=DEV)].cpu().numpy()])
    rq, rd = np.where(q > 0, remap[np.maximum(q, 0)], q), np.where(d > 0, remap[np.maximum(d, 0)], d)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if "embedding" not in k}
    want, wcounts, err = oracle.drmm(rq, rd, b["query_idf"][idx].cpu().numpy(), oracle.pack(small), 300, torch.linspace(-1, 1, 30)[1:].numpy(), "LCH", "IDF",
                                     sd["gates.weight"], small, sd["ffw.0.weight"], sd["ffw.0.bias"], sd["ffw.2.weight"], sd["ffw.2.bias"],
                                     sd["output_layer.weight"], sd["output_layer.bias"])
    assert err == 0
    assert np.array_equal(c0[idx].cpu().numpy(), wcounts)
    assert rel_err(s[idx].cpu().numpy(), want).max() <= ORACLE_TOL


def _repeated_term_docs(L, V, seed):
    """Documents that stress the distinct-term pass (interaction.h: distinct_terms): one term repeated over the whole document,
    terms that all land in one hash bucket (longest probe chains), heavy repetition with OOV terms and pads mixed in, no repetition."""
    rng = np.random.default_rng(seed)
    ids = np.arange(1, V, dtype=np.int64)
    bucket = ((ids * 2654435761) & 0xFFFFFFFF) >> 22
    same = ids[bucket == np.bincount(bucket).argmax()]          # every id of the fullest hash bucket
    d = np.zeros((6, L), dtype=np.int64)
    d[0, :] = 7
    d[1, :] = rng.choice(same, L)
    b0 = int(bucket[same[0] - 1])
    d[2, :] = rng.choice(ids[(bucket >= b0) & (bucket < b0 + 24)], L)   # ~470 ids over 24 adjacent buckets: one long cluster
    d[3, :] = rng.integers(1, 40, L)
    d[3, rng.random(L) < 0.2] = -1
    d[3, rng.random(L) < 0.2] = 0
    d[4, :] = rng.permutation(V - 1)[:L] + 1
    d[5, : L // 3] = rng.integers(1, V, L // 3)                 # short document, pads behind it
    q = np.array([[7, int(same[0]), 3, 0]] * 6, dtype=np.int64)
    return q, d


@pytest.mark.parametrize("L", [800, 896, 897, 1024, 5])
def test_repeated_document_terms(L):
    V, D = 20000, 300
    torch.manual_seed(L)
    rng = np.random.default_rng(L)
    emb = (rng.standard_normal((V, D)) * 0.4).astype(np.float32)
    emb[0] = 0
    q, d = _repeated_term_docs(L, V, L)
    idf = rng.random((6, 4)).astype(np.float32)
    packed = oracle.pack(emb)
    batch = {"query": _t(q), "posdoc": _t(d), "query_idf": _t(idf)}
    # KNRM: repeated terms are gathered once and weighted by their count; the oracle walks every position
    r = KNRM({}, SimpleNamespace(embeddings=emb))
    m = r.build_model().to(DEV).eval()
    with torch.no_grad():
        s = r.test(batch).cpu().numpy()
    mu, sigma = (x.cpu().numpy() for x in m.kernels.stacked())
    want, _ = oracle.knrm(q, d, packed, D, mu, sigma, m.combine[0].weight.detach().cpu().numpy(), m.combine[0].bias.detach().cpu().numpy())
    assert rel_err(s, want).max() <= ORACLE_TOL
    # DRMM: the bin counts stay bit-exact
    r = DRMM({}, SimpleNamespace(embeddings=emb))
    m = r.build_model().to(DEV).eval()
    with torch.no_grad():
        c0 = torch.empty((6, 4, 30), dtype=torch.int32, device=DEV)
        s = m(batch["posdoc"], batch["query"], batch["query_idf"], counts_out=c0).view(-1).cpu().numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if "embedding" not in k}
    want, wcounts, err = oracle.drmm(q, d, idf, packed, D, torch.linspace(-1, 1, 30)[1:].numpy(), "LCH", "IDF", sd["gates.weight"], emb,
                                     sd["ffw.0.weight"], sd["ffw.0.bias"], sd["ffw.2.weight"], sd["ffw.2.bias"], sd["output_layer.weight"],
                                     sd["output_layer.bias"])
    assert err == 0
    assert np.array_equal(c0.cpu().numpy(), wcounts)
    assert rel_err(s, want).max() <= ORACLE_TOL


@pytest.mark.parametrize("seed", range(40))
def test_random_geometries_against_the_oracle(seed):
    """Randomised sweep: document lengths 1..1100 (both sides of the distinct-term pass's 896-position limit), 1..12 query terms, embedding
    widths that fill 1..5 packed-row vectors, vocabularies small enough that most terms repeat, pads / OOV terms sprinkled in, a few
    all-pad documents - KNRM, DRMM (bin counts bit-exact) and DRMM-TKS against the C oracle."""
    from capreolus_amd.reranker import DRMMTKS

    torch.manual_seed(seed)          # (the models' initial weights)
    rng = np.random.default_rng(1000 + seed)
    L = int(rng.choice([1, 7, 40, 255, 256, 300, 800, 895, 896, 897, 1000, 1100]))
    Q = int(rng.integers(1, 13))
    D = int(rng.choice([20, 50, 64, 100, 128, 200, 300, 319]))
    V = int(rng.choice([8, 60, 500, 5000]))
    B = 24
    emb = (rng.standard_normal((V, D)) * 0.5).astype(np.float32)
    emb[0] = 0
    d = rng.integers(1, V, (B, L))
    d[rng.random((B, L)) < rng.choice([0.0, 0.1, 0.6])] = 0          # pads anywhere
    d[rng.random((B, L)) < rng.choice([0.0, 0.05])] = -1             # OOV terms
    lens = rng.integers(0, L + 1, B)
    d[np.arange(L)[None, :] >= lens[:, None]] = 0                    # padded tails (some documents are all pads)
    q = rng.integers(1, V, (B, Q))
    q[np.arange(Q)[None, :] >= rng.integers(1, Q + 1, B)[:, None]] = 0
    idf = rng.random((B, Q)).astype(np.float32)
    packed = oracle.pack(emb)
    batch = {"query": _t(q), "posdoc": _t(d), "query_idf": _t(idf)}

    r = KNRM({}, SimpleNamespace(embeddings=emb))
    m = r.build_model().to(DEV).eval()
    with torch.no_grad():
        got = r.test(batch).cpu().numpy()
    mu, sigma = (x.cpu().numpy() for x in m.kernels.stacked())
    w_c, b_c = m.combine[0].weight.detach().cpu().numpy(), m.combine[0].bias.detach().cpu().numpy()
    want, _ = oracle.knrm(q, d, packed, D, mu, sigma, w_c, b_c)
    # The score is a signed sum of 11 log-features: with random combine weights it can cancel to 1e-3 of its terms, and an element-wise
    # relative bound then measures the cancellation, not the kernel.  Bound the error by the terms instead (features through an identity
    # combine layer): 2e-5 of the score OR 1e-6 of sum |w_k f_k| + |b| (a dozen fp32 ulps of the largest term; observed < 1e-7).
    feats = np.stack([oracle.knrm(q, d, packed, D, mu, sigma, np.eye(len(mu), dtype=np.float32)[k:k + 1], np.zeros(1, np.float32))[0]
                      for k in range(len(mu))], 1)
    terms = np.abs(w_c.ravel()[None, :] * feats).sum(1) + abs(float(b_c[0]))
    err = np.abs(got - want)
    assert ((err <= ORACLE_TOL * np.abs(want)) | (err <= 1e-6 * terms)).all(), ("knrm", L, Q, D, V, float((err / terms).max()))

    r = DRMM({}, SimpleNamespace(embeddings=emb))
    m = r.build_model().to(DEV).eval()
    with torch.no_grad():
        c0 = torch.empty((B, Q, 30), dtype=torch.int32, device=DEV)
        got = m(batch["posdoc"], batch["query"], batch["query_idf"], counts_out=c0).view(-1).cpu().numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if "embedding" not in k}
    want, wcounts, err = oracle.drmm(q, d, idf, packed, D, torch.linspace(-1, 1, 30)[1:].numpy(), "LCH", "IDF", sd["gates.weight"], emb, sd["ffw.0.weight"],
                                     sd["ffw.0.bias"], sd["ffw.2.weight"], sd["ffw.2.bias"], sd["output_layer.weight"], sd["output_layer.bias"])
    assert err == 0
    assert np.array_equal(c0.cpu().numpy(), wcounts), ("drmm counts", L, Q, D, V)
    # (same conditioning argument: the score is out_w * x + out_b with |x| <= 1, and the two can cancel to 1e-4 of either)
    terms = np.abs(sd["output_layer.weight"]).sum() + np.abs(sd["output_layer.bias"]).sum()
    err = np.abs(got - want)
    assert ((err <= ORACLE_TOL * np.abs(want)) | (err <= 1e-6 * terms)).all(), ("drmm", L, Q, D, V, float(err.max() / terms))

    topk = int(min(L, rng.integers(1, 17)))
    r = DRMMTKS({"topk": topk}, SimpleNamespace(embeddings=emb))
    m = r.build_model().to(DEV).eval()
    with torch.no_grad():
        got = r.test(batch).cpu().numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if "embedding" not in k}
    want, err = oracle.drmmtks(q, d, idf, packed, D, topk, sd["gates.weight"], sd["ffw.0.weight"], sd["ffw.0.bias"], sd["output_layer.weight"],
                               sd["output_layer.bias"])
    assert err == 0
    terms = np.abs(sd["output_layer.weight"]).sum() + np.abs(sd["output_layer.bias"]).sum()
    err = np.abs(got - want)
    assert ((err <= ORACLE_TOL * np.abs(want)) | (err <= 1e-6 * terms)).all(), ("drmmtks", L, Q, D, V, topk, float(err.max() / terms))


@pytest.mark.parametrize("kind", ["knrm", "drmm"])
def test_ndcg20_parity_gpu_vs_reference(kind):
    from capreolus_amd import run_io
    from tests.helpers import run_from_scores, synthetic_qrels

    c = load_case(kind, "ranklist")
    r = _knrm_model(c) if kind == "knrm" else _drmm_model(c)
    with torch.no_grad():
        got = r.test(_batch(c)).cpu().numpy()
    diffs = []
    for seed in range(5):
        qrels = {"1": synthetic_qrels(len(got), seed)}
        ours = run_io.ndcg_cut(qrels, {"1": run_from_scores(got)}, 20)["1"]
        ref = run_io.ndcg_cut(qrels, {"1": run_from_scores(c["ref_scores"])}, 20)["1"]
        diffs.append(abs(ours - ref))
    if kind == "knrm":
        assert max(diffs) < 1e-12, diffs
        return
    # DRMM: the reference's own `(sim < 1.0).sum()` on cos(a, a) = 1 +- ulp moves bin counts on 78 of these 200 pairs (see
    # tests/test_oracle_golden.py::test_drmm_coin_flip_statistics), so against the REFERENCE the ranking agrees only within that noise
    # (measured: nDCG@20 delta <= 0.064 on these five qrel sets).  What must hold exactly is GPU == oracle: the two share their bin
    # counts bit for bit, and the oracle's back end reproduces the reference on the reference's counts (test_drmm_back_end_on_
    # reference_counts), so any difference between them would be a bug, not noise.
    assert max(diffs) <= 0.064 + 1e-9, diffs
    packed = oracle.pack(c["emb"])
    want, wcounts, err = oracle.drmm(c["query"], c["posdoc"], c["query_idf"], packed, int(c["D"]), c["edges"], str(c["histType"]), str(c["gateType"]),
                                     c["sd.gates.weight"], c["emb"], c["sd.ffw.0.weight"], c["sd.ffw.0.bias"], c["sd.ffw.2.weight"],
                                     c["sd.ffw.2.bias"], c["sd.output_layer.weight"], c["sd.output_layer.bias"])
    assert err == 0
    g16, w16 = got.astype(np.float16), want.astype(np.float16)
    assert np.array_equal(g16, w16), np.nonzero(g16 != w16)[0]                       # the fp16 scores predict() stores
    assert np.array_equal(rank_order(g16), rank_order(w16))
    for seed in range(5):
        qrels = {"1": synthetic_qrels(len(got), seed)}
        assert run_io.ndcg_cut(qrels, {"1": run_from_scores(got)}, 20)["1"] == run_io.ndcg_cut(qrels, {"1": run_from_scores(want)}, 20)["1"]


def test_predict_end_to_end_on_gpu(tmp_path):
    """PytorchTrainer.predict -> KNRM.test -> HIP kernel -> fp16 run file, against the oracle's ranking."""
    from capreolus_amd import run_io
    from capreolus_amd.trainer import PytorchTrainer

    c = load_case("knrm", "default")
    r = _knrm_model(c)
    B = c["query"].shape[0]

    class Sampler(torch.utils.data.IterableDataset):
        qid_to_docids = {"7": [f"d{i}" for i in range(B // 2)], "3": [f"d{i}" for i in range(B // 2, B)]}

        def __iter__(self):
            i = 0
            for qid, docs in self.qid_to_docids.items():
                for d in docs:
                    yield {"qid": qid, "posdocid": d, "query": c["query"][i], "posdoc": c["posdoc"][i], "query_idf": c["query_idf"][i]}
                    i += 1

        def __len__(self):
            return B

        def get_qid_docid_pairs(self):
            for qid, docs in self.qid_to_docids.items():
                for d in docs:
                    yield qid, d

    preds = PytorchTrainer({"batch": 5}).predict(r, Sampler(), tmp_path / "run.txt")
    flat = np.array([preds[q][d] for q, docs in Sampler.qid_to_docids.items() for d in docs], dtype=np.float32)
    assert rel_err(flat, c["ref_scores"].astype(np.float16).astype(np.float32)).max() < 2e-3  # fp16 grid
    run = run_io.load_trec_run(tmp_path / "run.txt")
    assert list(run.keys()) == ["3", "7"]  # qids in integer order (searcher/__init__.py:51)


def test_predict_at_the_reference_default_evalbatch_coalesces_launches():
    """evalbatch = batch = 32 (reference trainer/pytorch.py:24-25, 334) would be 32-workgroup launches: `predict` hands the kernel one
    batch per `coalesce` pairs instead.  Same predictions bit for bit as one scoring call per DataLoader batch; 1 launch, not 7."""
    from capreolus_amd.trainer import PytorchTrainer

    c = load_case("knrm", "ranklist")
    r = _knrm_model(c)
    B = c["query"].shape[0]

    class Sampler(torch.utils.data.IterableDataset):
        qid_to_docids = {"1": [f"d{i}" for i in range(B)]}

        def __iter__(self):
            for i in range(B):
                yield {"qid": "1", "posdocid": f"d{i}", "query": c["query"][i], "posdoc": c["posdoc"][i], "query_idf": c["query_idf"][i]}

        def __len__(self):
            return B

        def get_qid_docid_pairs(self):
            return (("1", f"d{i}") for i in range(B))

    calls = []
    orig = r.test
    r.test = lambda d: (calls.append(len(d["qid"])), orig(d))[1]
    per_batch = PytorchTrainer({"batch": 32, "coalesce": 0}).predict(r, Sampler())
    assert calls == [32] * 7                                # 200 pairs: six full batches + one filled by repetition
    calls.clear()
    merged = PytorchTrainer({"batch": 32}).predict(r, Sampler())
    assert calls == [B] and merged == per_batch
    got = np.array([merged["1"][f"d{i}"] for i in range(B)], dtype=np.float16)
    assert (got == c["ref_scores_f16"]).mean() > 0.98        # (fp16 rounding-boundary flips only, as test_knrm_rank_order)


def test_predict_over_rccl_single_rank(tmp_path):
    """The multi-GPU path of `predict` / `predict_resident` with the real collective: a "nccl" (= RCCL) process group of one rank,
    `all_gather_into_tensor` on device tensors, the padded `width` layout - same predictions as without a process group.  (The
    world_size 2 / 3 logic runs on gloo in tests/test_trainer_cpu.py; the 8-GPU run is the driver's.)"""
    import socket

    import torch.distributed as dist

    from capreolus_amd.feeder import CandidateStore
    from capreolus_amd.trainer import PytorchTrainer

    c = load_case("knrm", "default")
    r = _knrm_model(c)
    B = c["query"].shape[0]
    q2d = {"11": [f"d{i}" for i in range(B // 3)], "4": [f"d{i}" for i in range(B // 3, B)]}      # ragged lists

    class Sampler(torch.utils.data.IterableDataset):
        qid_to_docids = q2d

        def __iter__(self):
            i = 0
            for qid, docs in q2d.items():
                for d in docs:
                    yield {"qid": qid, "posdocid": d, "query": c["query"][i], "posdoc": c["posdoc"][i], "query_idf": c["query_idf"][i]}
                    i += 1

        def __len__(self):
            return B

        def get_qid_docid_pairs(self):
            return ((q, d) for q, docs in q2d.items() for d in docs)

    store = CandidateStore(DEV)
    i = 0
    for qid, docs in q2d.items():
        for d in docs:
            store.add_query(qid + ":" + d, c["query"][i], c["query_idf"][i])
            store.add_doc(qid + ":" + d, c["posdoc"][i])
            i += 1
    store.finalize()
    rq2d = {qid + ":" + d: [qid + ":" + d] for qid, docs in q2d.items() for d in docs}                # one (query row, doc row) per pair
    t = PytorchTrainer({"batch": 7})
    plain, plain_res = t.predict(r, Sampler()), t.predict_resident(r, store, rq2d)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        assert t.predict(r, Sampler(), tmp_path / "run.txt") == plain
        assert t.predict_resident(r, store, rq2d) == plain_res
        x = torch.arange(5, device=DEV, dtype=torch.float32)
        out = torch.empty(5, device=DEV)
        dist.all_gather_into_tensor(out, x)
        assert torch.equal(out, x)
    finally:
        dist.destroy_process_group()
    assert (tmp_path / "run.txt").exists()


@pytest.mark.parametrize("kind", ["knrm", "drmm"])
def test_resident_store_matches_extractor_layout(kind, tmp_path):
    """Row N1: int32 tables + index pairs give bit-identical scores to the int64 [B,Q]/[B,L] layout."""
    from capreolus_amd.feeder import CandidateStore
    from capreolus_amd.trainer import PytorchTrainer

    c = load_case(kind, "default")
    r = _knrm_model(c) if kind == "knrm" else _drmm_model(c)
    B = c["query"].shape[0]
    # two "queries" (first and second half of the fixture), documents shared between them where ids repeat
    q2d = {"11": [f"d{i}" for i in range(B // 2)], "4": [f"d{i}" for i in range(B // 2, B)]}
    row = {(q, d): i for i, (q, d) in enumerate((q, d) for q, ds in q2d.items() for d in ds)}
    store = CandidateStore(DEV)
    for (q, d), i in row.items():
        store.add_query(q + ":" + d, c["query"][i], c["query_idf"][i])  # every pair has its own query row in this fixture
        store.add_doc(d, c["posdoc"][i])
    store.finalize()
    pq = torch.arange(B, dtype=torch.int32, device=DEV)
    pd = torch.as_tensor([store.drow[d] for (_, d) in row], dtype=torch.int32, device=DEV)
    with torch.no_grad():
        want = r.test(_batch(c))
        got = r.test_resident(store, pq, pd)
        assert torch.equal(got, want)
        perm = torch.randperm(B, device=DEV)
        assert torch.equal(r.test_resident(store, pq[perm], pd[perm]), want[perm])


def test_predict_resident_ranklist():
    from capreolus_amd.feeder import CandidateStore
    from capreolus_amd.trainer import PytorchTrainer

    c = load_case("knrm", "ranklist")
    r = _knrm_model(c)
    B = c["query"].shape[0]
    q2d = {"301": [f"d{i}" for i in range(B)]}
    store = CandidateStore.from_id2vec(DEV, q2d, lambda q, d: {"query": c["query"][int(d[1:])], "posdoc": c["posdoc"][int(d[1:])],
                                                              "query_idf": c["query_idf"][int(d[1:])]})
    assert store.q_table.shape[0] == 1 and store.d_table.shape[0] == B  # the query is stored once for its 200 candidates
    preds = PytorchTrainer({"evalbatch": 64}).predict_resident(r, store, q2d)
    got = np.array([preds["301"][f"d{i}"] for i in range(B)], dtype=np.float16)
    assert (got == c["ref_scores_f16"]).mean() > 0.98


# ---- row N2: fp16 rounding + run-file order + nDCG@k on the device vs the host twins (run_io) --------------------------
def _ranking_case(seed, counts, coarse):
    rng = np.random.default_rng(seed)
    total = int(sum(counts))
    scores = rng.normal(size=total).astype(np.float32) * (3.0 if not coarse else 0.02)
    if coarse:  # few distinct fp16 values -> many ties
        scores = np.round(scores * 40) / 40
    if total > 12:
        scores[3] = 0.0
        scores[7] = -0.0
        scores[5] = 70000.0   # rounds to +inf in fp16
        scores[11] = 1e-8     # fp16 subnormal range -> 0
    q2d, k0 = {}, 0
    for qi, n in enumerate(counts):
        # docids whose string order differs from their list order
        q2d[str(100 - qi)] = [f"D{(i * 7919 + qi) % 100003:06d}" for i in range(n)]
        k0 += n
    return scores, q2d


@pytest.mark.parametrize("counts,coarse", [((1000,), False), ((1000, 37, 1, 0, 513), True), ((16384, 2), True), ((200,) * 64, True)])
def test_rank_candidates_matches_run_writer(counts, coarse, tmp_path):
    from capreolus_amd import ranking

    scores, q2d = _ranking_case(sum(counts), counts, coarse)
    off = ranking.offsets_of(counts, DEV)
    k = 1000
    idx, f16 = ranking.rank_candidates(torch.as_tensor(scores, device=DEV), off, k)
    engine.status_word(torch.device(DEV)).raise_if_set()
    idx, f16 = idx.cpu().numpy(), f16.cpu().numpy()
    # the reference's host path: astype(float16).item() into a dict, then write_trec_run's sort
    preds, pos = {}, 0
    for qid, docs in q2d.items():
        preds[qid] = {d: scores[pos + i].astype(np.float16).item() for i, d in enumerate(docs)}
        pos += len(docs)
    run_io.write_trec_run(preds, tmp_path / "run.txt")
    want = {}
    for line in open(tmp_path / "run.txt"):
        qid, _, docid, rank, score, _ = line.split()
        want.setdefault(qid, []).append((docid, float(score)))
    for qi, (qid, docs) in enumerate(q2d.items()):
        n = min(len(docs), k)
        got = [(docs[i], float(s)) for i, s in zip(idx[qi, :n], f16[qi, :n])]
        assert got == want.get(qid, [])[:n], qid
        assert (idx[qi, n:] == -1).all()


@pytest.mark.parametrize("counts,coarse", [((1000,) * 8, True), ((1000, 37, 1, 0, 513), False), ((5000, 20), True)])
def test_ndcg_cut_matches_host(counts, coarse):
    from capreolus_amd import ranking

    scores, q2d = _ranking_case(7 + sum(counts), counts, coarse)
    rng = np.random.default_rng(3)
    qrels = {}
    for qi, (qid, docs) in enumerate(q2d.items()):
        if qi == 2:
            continue  # a query without qrels
        qrels[qid] = {d: int(rng.integers(-1, 4)) for d in docs if rng.random() < 0.3}
        qrels[qid]["UNRETRIEVED"] = 3  # judged documents outside the candidate list count in the ideal ranking
    for k in (20, 10):
        rel, tie, idcg, off = ranking.eval_arrays(q2d, qrels, k, DEV)
        got = ranking.ndcg_cut(torch.as_tensor(scores, device=DEV), off, rel, tie, idcg, k=k).cpu().numpy()
        engine.status_word(torch.device(DEV)).raise_if_set()
        preds, pos = {}, 0
        for qid, docs in q2d.items():
            preds[qid] = {d: scores[pos + i].astype(np.float16).item() for i, d in enumerate(docs)}
            pos += len(docs)
        want = run_io.ndcg_cut(qrels, preds, k)
        for qi, qid in enumerate(q2d):
            if qid in want:
                assert abs(got[qi] - want[qid]) <= 1e-12, (qid, got[qi], want[qid])
            else:
                assert got[qi] == 0.0


def test_ndcg_cut_published_vectors_on_device():
    """capamd_ndcg_cut against the numbers pytrec_eval's README and the DCG article print (tests/helpers.py: NDCG_PUBLISHED)."""
    from capreolus_amd import ranking
    from tests.helpers import NDCG_PUBLISHED

    for source, qrels, run, k, want in NDCG_PUBLISHED:
        k = min(k, 256)             # the kernel's cut-off limit; every list here is shorter than that
        q2d = {qid: list(docs) for qid, docs in run.items()}
        scores = torch.tensor([s for docs in run.values() for s in docs.values()], dtype=torch.float32, device=DEV)
        rel, tie, idcg, off = ranking.eval_arrays(q2d, qrels, k, DEV)
        got = ranking.ndcg_cut(scores, off, rel, tie, idcg, k=k).cpu().numpy()
        engine.status_word(torch.device(DEV)).raise_if_set()
        for qi, qid in enumerate(q2d):
            v, tol = want[qid]
            assert abs(got[qi] - v) <= max(tol, 1e-12), (source, qid, got[qi], v)
    # score ties: docid descending, whatever the list order
    qrels = {"1": {"a": 1, "b": 0}}
    for docs in (["a", "b"], ["b", "a"]):
        rel, tie, idcg, off = ranking.eval_arrays({"1": docs}, qrels, 1, DEV)
        assert ranking.ndcg_cut(torch.ones(2, device=DEV), off, rel, tie, idcg, k=1).item() == 0.0


def test_ranking_flags_nan_and_bad_ties():
    from capreolus_amd import ranking

    s = torch.tensor([1.0, float("nan"), 0.5], device=DEV)
    off = ranking.offsets_of([3], DEV)
    idx, _ = ranking.rank_candidates(s, off, 3)
    assert idx.cpu().tolist() == [[0, 2, 1]]  # NaN ranked last
    with pytest.raises(ValueError):
        engine.status_word(torch.device(DEV)).raise_if_set()
    rel = torch.zeros(3, dtype=torch.int32, device=DEV)
    tie = torch.tensor([0, 1, 7], dtype=torch.int32, device=DEV)
    ranking.ndcg_cut(torch.tensor([1.0, 2.0, 3.0], device=DEV), off, rel, tie, torch.ones(1, dtype=torch.float64, device=DEV), k=2)
    with pytest.raises(ValueError):
        engine.status_word(torch.device(DEV)).raise_if_set()


def test_evaluate_resident_equals_host_pipeline():
    from capreolus_amd.feeder import CandidateStore
    from capreolus_amd.trainer import PytorchTrainer

    c = load_case("knrm", "ranklist")
    r = _knrm_model(c)
    B = c["query"].shape[0]
    q2d = {"301": [f"d{i}" for i in range(B // 2)], "302": [f"d{i}" for i in range(B // 2, B)]}
    store = CandidateStore.from_id2vec(DEV, q2d, lambda q, d: {"query": c["query"][int(d[1:])], "posdoc": c["posdoc"][int(d[1:])],
                                                              "query_idf": c["query_idf"][int(d[1:])]})
    rng = np.random.default_rng(0)
    qrels = {q: {d: int(rng.integers(0, 3)) for d in ds if rng.random() < 0.2} for q, ds in q2d.items()}
    tr = PytorchTrainer({"evalbatch": 64})
    got = tr.evaluate_resident(r, store, q2d, qrels, k=20)
    want = run_io.mean_ndcg_cut(qrels, tr.predict_resident(r, store, q2d), 20)
    assert abs(got - want) <= 1e-12
    assert tr.evaluate_resident(r, store, q2d, qrels, k=20) == got          # second call: the cached index pairs / judgment arrays
    qrels2 = {q: {d: 2 - g for d, g in ds.items()} for q, ds in qrels.items()}  # other judgments (another object): the plan is rebuilt
    got2 = tr.evaluate_resident(r, store, q2d, qrels2, k=20)
    assert abs(got2 - run_io.mean_ndcg_cut(qrels2, tr.predict_resident(r, store, q2d), 20)) <= 1e-12 and got2 != got
    assert abs(tr.evaluate_resident(r, store, q2d, qrels, k=10) - run_io.mean_ndcg_cut(qrels, tr.predict_resident(r, store, q2d), 10)) <= 1e-12


# ---- training step (row N3): HIP features + Jacobian diagonals vs autograd through the ATen port on CPU ----------
def test_knrm_training_gradients_match_autograd():
    from oracle import torch_port

    c = load_case("knrm", "twolayer_tanh")
    r = _knrm_model(c)
    m = r.model
    m.train()
    q, d = _t(c["query"]), _t(c["posdoc"])
    neg = d.roll(1, 0)
    pos_s, neg_s = r.score({"query": q, "posdoc": d, "negdoc": neg, "query_idf": _t(c["query_idf"])})
    loss = torch.clamp(1.0 - (pos_s - neg_s), min=0).mean() + 0.01 * pos_s.sum()
    loss.backward()
    # the same computation with plain ATen ops and autograd on the host
    emb = torch.as_tensor(c["emb"])
    mu, sigma, w1, b1, w2, b2 = (torch.as_tensor(x).clone().requires_grad_(True) for x in knrm_weights(c))
    qc, dc = torch.as_tensor(c["query"]), torch.as_tensor(c["posdoc"])
    ps = torch_port.knrm(emb, qc, dc, mu, sigma, w1, b1, w2, b2, bool(c["scoretanh"]))
    ns = torch_port.knrm(emb, qc, dc.roll(1, 0), mu, sigma, w1, b1, w2, b2, bool(c["scoretanh"]))
    ref_loss = torch.clamp(1.0 - (ps - ns), min=0).mean() + 0.01 * ps.sum()
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * max(1.0, abs(ref_loss.item()))
    got_mu = torch.stack([k.mu.grad for k in m.kernels.kernels]).cpu()
    got_sg = torch.stack([k.sigma.grad for k in m.kernels.kernels]).cpu()
    scale = lambda t: float(t.abs().max()) + 1e-8  # noqa: E731
    assert (got_mu - mu.grad).abs().max() <= 2e-3 * scale(mu.grad), (got_mu, mu.grad)
    assert (got_sg - sigma.grad).abs().max() <= 2e-3 * scale(sigma.grad), (got_sg, sigma.grad)
    assert (m.combine[0].weight.grad.cpu() - w1.grad).abs().max() <= 2e-3 * scale(w1.grad)
    assert (m.combine[2].weight.grad.cpu() - w2.grad).abs().max() <= 2e-3 * scale(w2.grad)


def test_drmm_training_step_matches_autograd():
    """Histogram features from the HIP kernel == the C oracle's counts (bit exact); loss and gradients of the tiny net on
    top of them == the same ATen ops under autograd on the host."""
    c = load_case("drmm", "zero_idf")
    r = _drmm_model(c)
    m = r.model
    m.train()
    b = _batch(c)
    negdoc = c["posdoc"][np.roll(np.arange(len(c["posdoc"])), 1)]
    s = r.score({**b, "negdoc": _t(negdoc)})
    loss = torch.clamp(1.0 - (s[0] - s[1]), min=0).mean() + 0.01 * s[0].sum()
    loss.backward()
    t = {k[3:]: torch.as_tensor(v).clone().requires_grad_(True) for k, v in c.items() if k.startswith("sd.")}
    packed = oracle.pack(c["emb"])

    def host_scores(doc):
        _, counts, err = oracle.drmm(c["query"], doc, c["query_idf"], packed, int(c["D"]), c["edges"], "CH", "IDF", c["sd.gates.weight"], c["emb"],
                                     c["sd.ffw.0.weight"], c["sd.ffw.0.bias"], c["sd.ffw.2.weight"], c["sd.ffw.2.bias"],
                                     c["sd.output_layer.weight"], c["sd.output_layer.bias"])
        assert err == 0
        feats = torch.log(torch.as_tensor(counts).float() + 1)                      # LCH (DRMM.py:71, :76)
        z = torch.tanh(torch.tanh(feats @ t["ffw.0.weight"].t() + t["ffw.0.bias"]) @ t["ffw.2.weight"].t() + t["ffw.2.bias"]).squeeze(-1)
        q = torch.as_tensor(c["query"])
        gl = torch.as_tensor(c["query_idf"]) * t["gates.weight"].view(-1)[0] + (q == 0).float() * -1e7
        return ((torch.softmax(gl, dim=1) * z).sum(1) * t["output_layer.weight"].view(-1)[0] + t["output_layer.bias"].view(-1)[0])

    ps, ns = host_scores(c["posdoc"]), host_scores(negdoc)
    ref = torch.clamp(1.0 - (ps - ns), min=0).mean() + 0.01 * ps.sum()
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-4 * max(1.0, abs(ref.item()))
    for name, mod in (("ffw.0.weight", m.ffw[0].weight), ("ffw.2.weight", m.ffw[2].weight), ("output_layer.weight", m.output_layer.weight)):
        g, gr = mod.grad.cpu(), t[name].grad
        assert (g - gr).abs().max() <= 2e-3 * (float(gr.abs().max()) + 1e-8), name


def test_train_loop_reduces_loss_and_checkpoints(tmp_path):
    """PytorchTrainer.train: a few iterations of pairwise training on a synthetic task where the positive document
    contains the query terms; the loss must go down, dev.best must be written and reloadable."""
    from capreolus_amd.trainer import PytorchTrainer

    rs = np.random.RandomState(5)
    V, L, Q = 600, 64, 4
    emb = synthetic.make_embeddings(V, 50, seed=9)
    r = KNRM({}, SimpleNamespace(embeddings=emb))
    r.build_model()
    r.model.combine[0].weight.data.zero_()  # every pair scores the same at the start -> hinge loss exactly 1
    queries = {str(q): rs.randint(1, V, size=Q) for q in range(12)}

    def doc(qid, relevant):
        d = rs.randint(1, V, size=L)
        if relevant:
            d[rs.choice(L, 6, replace=False)] = rs.choice(queries[qid], 6)
        return d

    class Train(torch.utils.data.IterableDataset):
        def __iter__(self):
            while True:
                qid = str(rs.randint(0, 12))
                yield {"query": torch.as_tensor(queries[qid]), "posdoc": torch.as_tensor(doc(qid, True)),
                       "negdoc": torch.as_tensor(doc(qid, False)), "query_idf": torch.zeros(Q)}

    docs = {(qid, f"d{i}"): doc(qid, i < 3) for qid in queries for i in range(10)}

    class Dev(torch.utils.data.IterableDataset):
        qid_to_docids = {qid: [f"d{i}" for i in range(10)] for qid in queries}

        def __iter__(self):
            for qid, ds in self.qid_to_docids.items():
                for d in ds:
                    yield {"qid": qid, "posdocid": d, "query": torch.as_tensor(queries[qid]), "posdoc": torch.as_tensor(docs[(qid, d)]),
                           "query_idf": torch.zeros(Q)}

        def __len__(self):
            return 120

        def get_qid_docid_pairs(self):
            for qid, ds in self.qid_to_docids.items():
                for d in ds:
                    yield qid, d

    qrels = {qid: {f"d{i}": int(i < 3) for i in range(10)} for qid in queries}
    t = PytorchTrainer({"batch": 16, "itersize": 64, "niters": 6, "lr": 0.02, "evalbatch": 40})
    losses = t.train(r, Train(), tmp_path / "train", Dev(), tmp_path / "dev", qrels, "ndcg_cut_20")
    assert t._use_fused and not t._fused_failed      # KNRM with a single-Linear combine trains through capamd_knrm_train_step (the plain Adam's state)
    assert not t.optimizer.param_groups[0].get("capturable") and not torch.is_tensor(t.optimizer.param_groups[0]["lr"])
    assert losses[0] > 0.1 and min(losses[1:]) < 0.2 * losses[0], losses  # starts at 1.0 per pair, separates within a few steps
    assert (tmp_path / "train" / "dev.best").exists() and (tmp_path / "dev" / "6.run").exists()
    before = r.model.combine[0].weight.detach().clone()
    r.model.combine[0].weight.data.zero_()
    t.load_best_model(r, tmp_path / "train")
    assert torch.isfinite(r.model.combine[0].weight).all() and not torch.equal(r.model.combine[0].weight, torch.zeros_like(before))


@pytest.mark.parametrize("kind", ["KNRM", "DRMM", "DRMMTKS", "PACRR", "ConvKNRM"])
def test_train_loop_runs_every_interaction_model_on_its_device_step(kind, tmp_path):
    """`PytorchTrainer.train` end to end for each trainable interaction model with its defaults: the trainer picks the reranker's
    device-kernel step (`capamd_*_train_step`, the plain Adam's state), the loss of a learnable synthetic task goes down, the dev set is
    predicted and ranked each iteration, `dev.best` is written and loads back."""
    import capreolus_amd.reranker as rr
    from capreolus_amd.trainer import PytorchTrainer

    rs = np.random.RandomState(11)
    V, L, Q = 600, 96, 4
    emb = synthetic.make_embeddings(V, 52, seed=9)
    torch.manual_seed(3)
    r = getattr(rr, kind)({}, SimpleNamespace(embeddings=emb, config={"maxqlen": Q}, pad=0))
    r.build_model()
    last = {"KNRM": "combine.0", "ConvKNRM": "combine.0", "DRMM": "output_layer", "DRMMTKS": "output_layer", "PACRR": "linear3"}[kind]
    r.model.get_submodule(last).weight.data.zero_()      # every pair scores the same at the start -> hinge loss exactly 1
    queries = {str(q): rs.randint(1, V, size=Q) for q in range(12)}
    idf = {qid: rs.uniform(0.5, 3.0, size=Q).astype(np.float32) for qid in queries}

    def doc(qid, relevant):
        d = rs.randint(1, V, size=L)
        d[rs.randint(L // 2, L):] = 0                       # padded tails of different lengths
        if relevant:
            d[rs.choice(L // 2, 8, replace=False)] = rs.choice(queries[qid], 8)
        return d

    class Train(torch.utils.data.IterableDataset):
        def __iter__(self):
            while True:
                qid = str(rs.randint(0, 12))
                yield {"query": torch.as_tensor(queries[qid]), "posdoc": torch.as_tensor(doc(qid, True)),
                       "negdoc": torch.as_tensor(doc(qid, False)), "query_idf": torch.as_tensor(idf[qid])}

    docs = {(qid, f"d{i}"): doc(qid, i < 3) for qid in queries for i in range(10)}

    class Dev(torch.utils.data.IterableDataset):
        qid_to_docids = {qid: [f"d{i}" for i in range(10)] for qid in queries}

        def __iter__(self):
            for qid, ds in self.qid_to_docids.items():
                for d in ds:
                    yield {"qid": qid, "posdocid": d, "query": torch.as_tensor(queries[qid]), "posdoc": torch.as_tensor(docs[(qid, d)]),
                           "query_idf": torch.as_tensor(idf[qid])}

        def __len__(self):
            return 120

        def get_qid_docid_pairs(self):
            for qid, ds in self.qid_to_docids.items():
                for d in ds:
                    yield qid, d

    qrels = {qid: {f"d{i}": int(i < 3) for i in range(10)} for qid in queries}
    t = PytorchTrainer({"batch": 16, "itersize": 64, "niters": 8, "lr": 0.01, "evalbatch": 40})
    losses = t.train(r, Train(), tmp_path / "train", Dev(), tmp_path / "dev", qrels, "ndcg_cut_20")
    assert t._use_fused and not t._fused_failed, kind
    assert not t.optimizer.param_groups[0].get("capturable")
    # (a loss per ITERATION of four batches: it starts at 1.0 per pair and the first iteration's mean is already below that)
    assert all(np.isfinite(losses)) and losses[0] > 0.05 and min(losses[1:]) < 0.9 * losses[0], (kind, losses)
    assert (tmp_path / "train" / "dev.best").exists() and (tmp_path / "dev" / "8.run").exists()
    t.load_best_model(r, tmp_path / "train")
    preds = t.predict(r, Dev())
    assert len(preds) == 12 and all(len(v) == 10 and all(np.isfinite(list(v.values()))) for v in preds.values())


@pytest.mark.parametrize("name", ["default", "top3_short", "ranklist"])
def test_drmmtks_scores(name):
    from capreolus_amd.reranker import DRMMTKS

    c = load_case("drmmtks", name)
    r = DRMMTKS({"topk": int(c["topk"])}, SimpleNamespace(embeddings=c["emb"]))
    m = r.build_model()
    m.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in c.items() if k.startswith("sd.")}, strict=False)
    m.to(DEV).eval()
    with torch.no_grad():
        got = r.test(_batch(c)).cpu().numpy()
    want, err = oracle.drmmtks(c["query"], c["posdoc"], c["query_idf"], oracle.pack(c["emb"]), int(c["D"]), int(c["topk"]), c["sd.gates.weight"],
                               c["sd.ffw.0.weight"], c["sd.ffw.0.bias"], c["sd.output_layer.weight"], c["sd.output_layer.bias"])
    assert err == 0
    assert rel_err(got, want).max() <= ORACLE_TOL, rel_err(got, want).max()
    assert rel_err(got, c["ref_scores"]).max() <= REL_TOL, rel_err(got, c["ref_scores"]).max()
    if name == "ranklist":
        same = got.astype(np.float16) == c["ref_scores_f16"]
        assert same.mean() > 0.98


@pytest.mark.parametrize("L,topk", [(800, 10), (896, 16), (897, 3), (40, 16)])
def test_drmmtks_repeated_document_terms(L, topk):
    """The distinct-term pass in front of DRMM-TKS: a repeated term is gathered once and enters the top-k lists `count` times (capped at
    k) - documents that are one term throughout, hash-bucket clusters, OOV / pad mixes, no repeats; both sides of the 896-position limit."""
    from capreolus_amd.reranker import DRMMTKS

    V, D = 20000, 300
    torch.manual_seed(L + topk)
    rng = np.random.default_rng(L + topk)
    emb = (rng.standard_normal((V, D)) * 0.4).astype(np.float32)
    emb[0] = 0
    q, d = _repeated_term_docs(L, V, L + topk)
    idf = rng.random((6, 4)).astype(np.float32)
    r = DRMMTKS({"topk": topk}, SimpleNamespace(embeddings=emb))
    m = r.build_model().to(DEV).eval()
    with torch.no_grad():
        got = r.test({"query": _t(q), "posdoc": _t(d), "query_idf": _t(idf)}).cpu().numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if "embedding" not in k}
    want, err = oracle.drmmtks(q, d, idf, oracle.pack(emb), D, topk, sd["gates.weight"], sd["ffw.0.weight"], sd["ffw.0.bias"],
                               sd["output_layer.weight"], sd["output_layer.bias"])
    assert err == 0
    assert rel_err(got, want).max() <= ORACLE_TOL, rel_err(got, want).max()


def _pacrr_reranker(c):
    from capreolus_amd.reranker import PACRR

    cfg = {k: int(c[f"cfg.{k}"]) for k in ("mingram", "maxgram", "nfilters", "kmax", "combine")}
    cfg.update(idf=bool(int(c["cfg.idf"])), nonlinearity=str(c["nonlinearity"]))
    r = PACRR(cfg, SimpleNamespace(embeddings=c["emb"], config={"maxqlen": c["query"].shape[1]}))
    m = r.build_model()
    missing, unexpected = m.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in c.items() if k.startswith("sd.")}, strict=False)
    assert not unexpected and set(missing) <= {"embedding.weight"}, (missing, unexpected)   # the reference's parameter names, all of them
    m.to(DEV).eval()
    return r


@pytest.mark.parametrize("valu", ["0", "1"])
@pytest.mark.parametrize("name", PACRR_CASES)
def test_pacrr_scores(name, valu, monkeypatch):
    monkeypatch.setenv("CAPAMD_PACRR_VALU", valu)           # "1": the general fp32-VALU kernel instead of the matrix-pipe one
    c = load_case("pacrr", name)
    r = _pacrr_reranker(c)
    with torch.no_grad():
        got = r.test(_batch(c)).cpu().numpy()
    want, err = oracle.pacrr(c["query"], c["posdoc"], c["query_idf"], oracle.pack(c["emb"]), int(c["D"]), *pacrr_args(c))
    assert err == 0
    assert rel_err(got, want).max() <= ORACLE_TOL, rel_err(got, want).max()
    assert rel_err(got, c["ref_scores"]).max() <= REL_TOL, rel_err(got, c["ref_scores"]).max()
    if name == "ranklist":
        assert (got.astype(np.float16) == c["ref_scores_f16"]).mean() > 0.98


@pytest.mark.parametrize("Q,L,lo,hi,nf,kmax,idf,nonlin,comb", [
    (1, 1, 1, 1, 1, 1, False, "none", 1),        # smallest everything
    (8, 800, 1, 3, 32, 4, True, "relu", 128),    # the kernel's compile-time limits
    (5, 37, 2, 3, 7, 3, True, "tanh", 9),        # no unigram module, odd sizes
    (3, 256, 1, 2, 16, 2, False, "relu", 32),    # run length exactly 4
    (4, 1000, 1, 3, 8, 2, True, "relu", 16),     # longest supported document
    (2, 2, 3, 3, 4, 2, False, "relu", 4),        # window larger than the matrix: only padding contributes beyond (0,0)
    (5, 800, 1, 3, 32, 4, True, "relu", 32),     # the matrix-pipe kernel at its limits (Q = 5, 32 filters)
    (5, 64, 1, 3, 32, 2, False, "relu", 32),     # exactly one 64-position step
    (4, 65, 3, 3, 32, 2, False, "relu", 32),     # one position into the second step, trigrams only
    (6, 300, 1, 3, 32, 2, True, "relu", 32),     # one query row too many for it: general kernel
    (4, 300, 1, 3, 33, 2, True, "relu", 32),     # one filter too many for it: general kernel
    (5, 896, 1, 3, 32, 4, True, "relu", 32),     # the longest document the distinct-term pass hashes (150-word vocabulary: every term repeats)
    (4, 897, 1, 3, 32, 2, False, "relu", 32),    # one position more: every position listed on its own
    (4, 320, 1, 3, 32, 2, True, "relu", 32),     # the shortest document whose matrix planes have room for the hash
    (4, 288, 1, 3, 32, 2, True, "relu", 32),     # ... and one step below
])
def test_pacrr_geometries_match_oracle(Q, L, lo, hi, nf, kmax, idf, nonlin, comb):
    rng = np.random.default_rng(Q * 1000 + L)
    V, D, B = 150, 60, 19
    emb = synthetic.make_embeddings(V, D, seed=5)
    q = rng.integers(0, V, (B, Q)); d = rng.integers(0, V, (B, L))
    q[:, Q - 1:] *= rng.integers(0, 2, (B, 1)); d[:, L // 2:] *= rng.integers(0, 2, (B, 1))     # padded tails
    if B > 3:
        d[3] = 0                                                                                # all-padding document
    idfv = rng.random((B, Q), dtype=np.float32) * 6
    n = hi - lo + 1
    cws = [rng.standard_normal((nf, 1, g, g)).astype(np.float32) * 0.5 for g in range(lo, hi + 1)]
    cbs = [rng.standard_normal(nf).astype(np.float32) * 0.1 for _ in range(n)]
    F = Q * (n * kmax + int(idf))
    w1 = rng.standard_normal((comb, F)).astype(np.float32) * 0.3; b1 = rng.standard_normal(comb).astype(np.float32) * 0.1
    w2 = rng.standard_normal((comb, comb)).astype(np.float32) * 0.3; b2 = rng.standard_normal(comb).astype(np.float32) * 0.1
    w3 = rng.standard_normal((1, comb)).astype(np.float32) * 0.3; b3 = rng.standard_normal(1).astype(np.float32)
    want, err = oracle.pacrr(q, d, idfv, oracle.pack(emb), D, lo, hi, nf, kmax, cws, cbs, idf, w1, b1, w2, b2, w3, b3, nonlin)
    assert err == 0
    pe = engine.PackedEmbedding()
    got = engine.pacrr_forward(_t(q), _t(d), _t(idfv), pe.get(_t(emb)), V, D, lo, hi, nf, kmax, _t(np.concatenate([w.ravel() for w in cws])),
                               _t(np.concatenate(cbs)), idf, nonlin, _t(w1), _t(b1), _t(w2), _t(b2), _t(w3.ravel()), _t(b3)).cpu().numpy()
    assert rel_err(got, want).max() <= ORACLE_TOL, rel_err(got, want).max()


@pytest.mark.parametrize("Q", [4, 5])
@pytest.mark.parametrize("last", [-1, 0, 62, 63, 64, 126, 127, 128, 191, 199])
def test_pacrr_positions_behind_the_last_term_count_as_the_reference_counts_them(last, Q, monkeypatch):
    """The matrix-pipe kernels (the Q <= 4 form and the Q = 5 form) stop their convolutions one 64-position step behind the document's last term: every window further
    on is all padding and has the same value, and the k-max can use at most kmax copies of it (PACRR.py:64-75 takes the k largest over ALL
    positions).  Documents whose last term sits on and around the step boundaries, with padding INSIDE the document and a bias vector that
    makes the padding value the largest (positive biases) or irrelevant (negative ones): the oracle's scores, the general kernel's, and the
    whole-list route's bit for bit."""
    L, lo, hi, nf, kmax, comb, V, D = 200, 1, 3, 32, 4, 16, 150, 60
    rng = np.random.default_rng(1000 + last + 17 * Q)
    emb = synthetic.make_embeddings(V, D, seed=5)
    B = 12
    q = rng.integers(1, V, (B, Q)); d = rng.integers(1, V, (B, L))
    d[:, last + 1:] = 0
    if last > 3:
        d[:, 2:last:3] *= rng.integers(0, 2, (B, len(range(2, last, 3))))       # padding inside the document
        d[:, last] = rng.integers(1, V, B)
    idfv = rng.random((B, Q), dtype=np.float32) * 6
    n = hi - lo + 1
    cws = [rng.standard_normal((nf, 1, g, g)).astype(np.float32) * 0.5 for g in range(lo, hi + 1)]
    sign = 1.0 if last % 2 == 0 else -1.0
    cbs = [(sign * np.abs(rng.standard_normal(nf)) * 0.6).astype(np.float32) for _ in range(n)]
    F = Q * (n * kmax + 1)
    w1 = rng.standard_normal((comb, F)).astype(np.float32) * 0.3; b1 = rng.standard_normal(comb).astype(np.float32) * 0.1
    w2 = rng.standard_normal((comb, comb)).astype(np.float32) * 0.3; b2 = rng.standard_normal(comb).astype(np.float32) * 0.1
    w3 = rng.standard_normal((1, comb)).astype(np.float32) * 0.3; b3 = rng.standard_normal(1).astype(np.float32)
    want, err = oracle.pacrr(q, d, idfv, oracle.pack(emb), D, lo, hi, nf, kmax, cws, cbs, True, w1, b1, w2, b2, w3, b3, "relu")
    assert err == 0
    pe = engine.PackedEmbedding()
    args = (pe.get(_t(emb)), V, D, lo, hi, nf, kmax, _t(np.concatenate([w.ravel() for w in cws])), _t(np.concatenate(cbs)), True, "relu",
            _t(w1), _t(b1), _t(w2), _t(b2), _t(w3.ravel()), _t(b3))
    got = engine.pacrr_forward(_t(q), _t(d), _t(idfv), *args).cpu().numpy()
    # (against the batch's largest score: with random combine weights a score can be a cancellation to ~0, where 2e-7 is not 2e-5 of it)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= ORACLE_TOL * scale, np.abs(got - want).max() / scale
    monkeypatch.setenv("CAPAMD_PACRR_VALU", "1")
    valu = engine.pacrr_forward(_t(q), _t(d), _t(idfv), *args).cpu().numpy()
    assert np.abs(valu - got).max() <= ORACLE_TOL * scale
    monkeypatch.delenv("CAPAMD_PACRR_VALU")
    if Q > 4:
        return                 # (the whole-list route takes up to four query terms)
    ql = np.repeat(q[:3], 4, axis=0); il = np.repeat(idfv[:3], 4, axis=0)            # three lists of four documents
    pair = engine.pacrr_forward(_t(ql), _t(d), _t(il), *args)
    lists = engine.pacrr_forward_lists(np.array([0, 4, 8, 12]), _t(il), *args, query=_t(ql), doc=_t(d))
    assert torch.equal(pair, lists)
    # ... and with a workspace that has no room for the pairs' features: the combine layers inside the convolution kernel, the same bits
    lists2 = engine.pacrr_forward_lists(np.array([0, 4, 8, 12]), _t(il), *args, query=_t(ql), doc=_t(d), pair_part=False)
    assert torch.equal(pair, lists2)


@pytest.mark.parametrize("comb,kmax,idf", [(8, 2, True), (32, 2, True), (32, 4, True), (48, 2, False), (100, 3, True), (128, 4, True)])
def test_pacrr_list_route_combine_layers_in_one_pass_give_the_per_pair_bits(comb, kmax, idf):
    """The whole-list route runs the combine layers of all pairs behind the convolutions (pacrr_head32_lists_kernel for the default head
    size - inputs and width <= 32 -, pacrr_head_lists_kernel<32 / 64 / 128> otherwise, every pair's own head inside the convolution kernel
    when the weights do not fit a workgroup's LDS or the workspace has no per-pair part): each form against the per-pair kernel, bit for bit."""
    Q, L, lo, hi, nf, V, D, B = 4, 120, 1, 3, 32, 150, 60, 40
    rng = np.random.default_rng(comb * 10 + kmax)
    emb = synthetic.make_embeddings(V, D, seed=5)
    q = np.repeat(rng.integers(1, V, (4, Q)), 10, axis=0); d = rng.integers(0, V, (B, L))
    d[:, 70:] *= rng.integers(0, 2, (B, 1))
    idfv = np.repeat(rng.random((4, Q), dtype=np.float32) * 6, 10, axis=0)
    n = hi - lo + 1
    cws = [rng.standard_normal((nf, 1, g, g)).astype(np.float32) * 0.5 for g in range(lo, hi + 1)]
    cbs = [rng.standard_normal(nf).astype(np.float32) * 0.1 for _ in range(n)]
    F = Q * (n * kmax + int(idf))
    w1 = rng.standard_normal((comb, F)).astype(np.float32) * 0.3; b1 = rng.standard_normal(comb).astype(np.float32) * 0.1
    w2 = rng.standard_normal((comb, comb)).astype(np.float32) * 0.2; b2 = rng.standard_normal(comb).astype(np.float32) * 0.1
    w3 = rng.standard_normal((1, comb)).astype(np.float32) * 0.3; b3 = rng.standard_normal(1).astype(np.float32)
    pe = engine.PackedEmbedding()
    args = (pe.get(_t(emb)), V, D, lo, hi, nf, kmax, _t(np.concatenate([w.ravel() for w in cws])), _t(np.concatenate(cbs)), idf, "tanh",
            _t(w1), _t(b1), _t(w2), _t(b2), _t(w3.ravel()), _t(b3))
    pair = engine.pacrr_forward(_t(q), _t(d), _t(idfv), *args)
    want, err = oracle.pacrr(q, d, idfv, oracle.pack(emb), D, lo, hi, nf, kmax, cws, cbs, idf, w1, b1, w2, b2, w3, b3, "tanh")
    assert err == 0 and np.abs(pair.cpu().numpy() - want).max() <= ORACLE_TOL * np.abs(want).max()
    off = np.array([0, 10, 20, 30, 40])
    assert torch.equal(pair, engine.pacrr_forward_lists(off, _t(idfv), *args, query=_t(q), doc=_t(d)))
    assert torch.equal(pair, engine.pacrr_forward_lists(off, _t(idfv), *args, query=_t(q), doc=_t(d), pair_part=False))


def test_pacrr_errors():
    c = load_case("pacrr", "tanh_noidf_short")
    r = _pacrr_reranker(c)
    b = _batch(c)
    bad = dict(b, posdoc=b["posdoc"].clone())
    bad["posdoc"][2, 5] = int(c["V"]) + 3
    with pytest.raises(IndexError):
        r.test(bad)
    with pytest.raises(RuntimeError):                       # torch.topk's error for k > L (PACRR.py:74)
        r.test(dict(b, posdoc=b["posdoc"][:, :1]))
    pe = engine.PackedEmbedding()
    with pytest.raises(EngineError):                        # beyond the compile-time limits: refused, never truncated
        z = torch.zeros(4096, device=DEV)
        engine.pacrr_forward(b["query"], b["posdoc"], b["query_idf"], pe.get(_t(c["emb"])), int(c["V"]), int(c["D"]), 1, 4, 8, 2, z, z, False,
                             "relu", z, z, z, z, z, z)


CONVKNRM_ORACLE_TOL = 5e-5   # the oracle accumulates in double; the kernel in fp32 (two-term f16 split on the matrix pipe, ~2^-22)


def _convknrm_reranker(c):
    from capreolus_amd.reranker import ConvKNRM

    cfg = {k: int(c[f"cfg.{k}"]) for k in ("maxngram", "filters")}
    cfg.update({k: bool(int(c[f"cfg.{k}"])) for k in ("gradkernels", "crossmatch", "scoretanh", "singlefc")})
    r = ConvKNRM(cfg, SimpleNamespace(embeddings=c["emb"], pad=0))
    m = r.build_model()
    sd = {k[3:]: torch.as_tensor(v) for k, v in c.items() if k.startswith("sd.")}
    ws, bs = convknrm_conv_weights(int(c["conv_seed"]), cfg["filters"], int(c["D"]), cfg["maxngram"])
    for g, (w, b) in enumerate(zip(ws, bs)):
        sd[f"convs.{g}.0.weight"], sd[f"convs.{g}.0.bias"] = torch.as_tensor(w), torch.as_tensor(b)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {"embeddings.weight"}, (missing, unexpected)   # the reference's parameter names, all of them
    m.to(DEV).eval()
    return r


@pytest.mark.parametrize("name", CONVKNRM_CASES)
def test_convknrm_scores(name):
    c = load_case("convknrm", name)
    r = _convknrm_reranker(c)
    with torch.no_grad():
        got = r.test(_batch(c)).cpu().numpy()
    want, err = oracle.convknrm(c["query"], c["posdoc"], c["emb"], *convknrm_args(c))
    assert err == 0
    assert rel_err(got, want).max() <= CONVKNRM_ORACLE_TOL, rel_err(got, want).max()
    assert rel_err(got, c["ref_scores"]).max() <= REL_TOL, rel_err(got, c["ref_scores"]).max()
    if name == "ranklist":
        assert (got.astype(np.float16) == c["ref_scores_f16"]).mean() > 0.98


@pytest.mark.parametrize("name", CONVKNRM_CASES)
def test_convknrm_training_step(name):
    """ConvKNRM.score() in train mode (autograd through the convolutions, ConvKNRM.py:42-77): its scores reproduce the REFERENCE's on
    the fixtures (1e-5) and the fused scoring kernel's; the loss reaches the convolutions, the kernels and the combine layer, and one
    Adam step on a pairwise hinge loss lowers it."""
    c = load_case("convknrm", name)
    r = _convknrm_reranker(c)
    m = r.model
    b = _batch(c)
    with torch.no_grad():
        fused = r.test(b)
    m.train()
    pos, neg = r.score({**b, "negdoc": torch.roll(b["posdoc"], 1, 0)})
    assert pos.requires_grad
    assert rel_err(pos.detach().cpu().numpy(), c["ref_scores"]).max() <= 1e-5
    assert rel_err(pos.detach().cpu().numpy(), fused.cpu().numpy()).max() <= 5e-5
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3)
    loss0 = torch.clamp(1.0 - (pos - neg), min=0).mean()
    loss0.backward()
    for g in range(int(c["cfg.maxngram"])):
        assert m.convs[g][0].weight.grad is not None and float(m.convs[g][0].weight.grad.abs().max()) > 0
    assert m.combine[0].weight.grad is not None and m.embeddings.weight.grad is None
    if bool(int(c["cfg.gradkernels"])):
        assert m.kernels.kernels[3].mu.grad is not None
    opt.step()
    opt.zero_grad()
    pos, neg = r.score({**b, "negdoc": torch.roll(b["posdoc"], 1, 0)})
    assert float(torch.clamp(1.0 - (pos - neg), min=0).mean().detach()) < float(loss0.detach())
    with pytest.raises(RuntimeError):   # no CPU path, in training either
        m.cpu()
        r.score({k: v.cpu() for k, v in {**b, "negdoc": b["posdoc"]}.items()})


def test_convknrm_tables_match_direct_projection():
    rng = np.random.default_rng(3)
    V, D, F, G = 123, 77, 48, 3
    emb = rng.standard_normal((V, D)).astype(np.float32)
    ws = [rng.standard_normal((F, D, g)).astype(np.float32) for g in range(1, G + 1)]
    bs = [rng.standard_normal(F).astype(np.float32) for _ in range(G)]
    t = engine.ConvProjectionTables().get(_t(emb), [_t(w) for w in ws], [_t(b) for b in bs]).cpu().numpy().reshape(V, 6, F)
    order = [(1, 0), (2, 0), (3, 0), (2, 1), (3, 1), (3, 2)]          # (n-gram size, tap) of the six parts
    for p, (g, tap) in enumerate(order):
        want = emb.astype(np.float64) @ ws[g - 1][:, :, tap].T.astype(np.float64) + (bs[g - 1] if tap == 0 else 0.0)
        assert np.abs(t[:, p] - want).max() <= 1e-4 * np.abs(want).max(), (p, np.abs(t[:, p] - want).max())


@pytest.mark.parametrize("Q,L,G,F,cross,single,tanh", [
    (1, 1, 1, 16, True, True, False),       # smallest everything
    (8, 300, 3, 128, True, True, False),    # 72 (view, query term) rows
    (5, 33, 2, 64, False, False, True),     # one position into the second tile
    (3, 1000, 3, 96, True, False, False),   # filters not a power of two
    (4, 64, 3, 32, False, True, False),     # exactly two tiles when every token is real
])
def test_convknrm_geometries_match_oracle(Q, L, G, F, cross, single, tanh):
    rng = np.random.default_rng(Q * 1000 + L)
    V, D, B = 150, 40, 13
    emb = synthetic.make_embeddings(V, D, seed=9)
    emb[0] = rng.standard_normal(D) * 0.1                     # a pad row that is NOT zero: it still feeds the neighbours' n-grams
    q = rng.integers(1, V, (B, Q)); d = rng.integers(1, V, (B, L))
    q[:, Q - 1:] *= rng.integers(0, 2, (B, 1)); d[:, L // 2:] *= rng.integers(0, 2, (B, 1))
    d[:, ::7] *= rng.integers(0, 2, (B, d[:, ::7].shape[1]))   # pads inside the document
    if L > 1:
        d[3] = 0                                              # all-padding document
        q[5] = 0                                              # all-padding query
    ws, bs = convknrm_conv_weights(17, F, D, G)
    K = 11
    views = G * G if cross else G
    mu = np.array([-0.9, -0.7, -0.5, -0.3, -0.1, 0.1, 0.3, 0.5, 0.7, 0.9, 1.0], dtype=np.float32)
    sigma = np.array([0.1] * 10 + [0.001], dtype=np.float32)
    H = 0 if single else 30
    w1 = (rng.standard_normal((max(H, 1), K * views)) * 0.2).astype(np.float32); b1 = rng.standard_normal(max(H, 1)).astype(np.float32) * 0.1
    w2 = None if single else (rng.standard_normal((1, H)) * 0.3).astype(np.float32)
    b2 = None if single else rng.standard_normal(1).astype(np.float32)
    want, err = oracle.convknrm(q, d, emb, ws, bs, cross, mu, sigma, w1, b1, w2, b2, tanh)
    assert err == 0
    tables = engine.ConvProjectionTables().get(_t(emb), [_t(w) for w in ws], [_t(b) for b in bs])
    got = engine.convknrm_forward(_t(q), _t(d), tables, V, G, F, cross, _t(mu), _t(sigma), _t(w1), _t(b1), None if single else _t(w2.ravel()),
                                  None if single else _t(b2), tanh).cpu().numpy()
    assert rel_err(got, want).max() <= CONVKNRM_ORACLE_TOL, rel_err(got, want).max()


def test_convknrm_errors():
    c = load_case("convknrm", "nocross_2fc_short")
    r = _convknrm_reranker(c)
    b = _batch(c)
    for bad_id in (int(c["V"]) + 3, -2):                      # nn.Embedding raises IndexError for both
        bad = dict(b, posdoc=b["posdoc"].clone())
        bad["posdoc"][2, 5] = bad_id
        with pytest.raises(IndexError):
            r.test(bad)
    with pytest.raises(ValueError):                           # filters not a multiple of 16: refused when the tables are built
        engine.ConvProjectionTables().get(_t(c["emb"]), [torch.zeros((20, int(c["D"]), 1), device=DEV)], [torch.zeros(20, device=DEV)])


@pytest.mark.parametrize("kind", ["drmmtks", "pacrr", "convknrm"])
def test_full_size_sibling_properties(full, kind):
    """Row-N4 models at the benchmark's geometry (vocabulary 400,001 x 300, 800-term documents; ConvKNRM: the 1.2 GB projection table):
    pairs are independent (a permutation of the batch permutes the scores bit for bit), the launch size does not matter (chunked
    calls == one call, bit for bit), and a sample of pairs agrees with the oracle."""
    from capreolus_amd.reranker import DRMMTKS, PACRR, ConvKNRM

    emb, batch = full
    if kind == "convknrm":
        batch = {k: (v.abs() if v.dtype == torch.int64 else v) for k, v in batch.items()}
    torch.manual_seed(3)
    ext = SimpleNamespace(embeddings=np.zeros((2, 300), dtype=np.float32), config={"maxqlen": 4}, pad=0)
    r = {"drmmtks": DRMMTKS, "pacrr": PACRR, "convknrm": ConvKNRM}[kind]({}, ext)
    m = r.build_model().to(DEV).eval()
    setattr(m, "embeddings" if kind == "convknrm" else "embedding", torch.nn.Embedding.from_pretrained(emb, freeze=True))
    with torch.no_grad():
        s = r.test(batch)
        assert s.shape == (2000,) and torch.isfinite(s).all()
        perm = torch.randperm(2000, device=DEV)
        assert torch.equal(r.test({k: v[perm] for k, v in batch.items()}), s[perm])
        sc = torch.cat([r.test({k: v[i:i + 333] for k, v in batch.items()}) for i in range(0, 2000, 333)])
        assert torch.equal(sc, s)
    idx = np.arange(0, 2000, 125)
    q, d, idf = (batch[k][idx].cpu().numpy() for k in ("query", "posdoc", "query_idf"))
    used = np.unique(np.concatenate([q.ravel(), d.ravel()]))
    used = used[used > 0]
    remap = np.zeros(400001, dtype=np.int64)
    remap[used] = np.arange(1, len(used) + 1)
    small = np.concatenate([np.zeros((1, 300), np.float32), emb[torch.as_tensor(used, device=DEV)].cpu().numpy()])
    rq, rd = np.where(q > 0, remap[np.maximum(q, 0)], q), np.where(d > 0, remap[np.maximum(d, 0)], d)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if "embedding" not in k}
    if kind == "drmmtks":
        want, err = oracle.drmmtks(rq, rd, idf, oracle.pack(small), 300, m.topk, sd["gates.weight"], sd["ffw.0.weight"], sd["ffw.0.bias"],
                                   sd["output_layer.weight"], sd["output_layer.bias"])
        tol = ORACLE_TOL
    elif kind == "pacrr":
        p = m.p
        want, err = oracle.pacrr(rq, rd, idf, oracle.pack(small), 300, p["mingram"], p["maxgram"], p["nfilters"], p["kmax"],
                                 [sd[f"ngrams.{i}.conv.weight"] for i in range(3)], [sd[f"ngrams.{i}.conv.bias"] for i in range(3)], p["idf"],
                                 sd["linear1.weight"], sd["linear1.bias"], sd["linear2.weight"], sd["linear2.bias"], sd["linear3.weight"],
                                 sd["linear3.bias"], p["nonlinearity"])
        tol = ORACLE_TOL
    else:
        mu, sigma = (x.cpu().numpy() for x in m.kernels.stacked())
        want, err = oracle.convknrm(rq, rd, small, [sd[f"convs.{i}.0.weight"] for i in range(3)], [sd[f"convs.{i}.0.bias"] for i in range(3)], True,
                                    mu, sigma, sd["combine.0.weight"], sd["combine.0.bias"])
        tol = CONVKNRM_ORACLE_TOL
    assert err == 0
    assert rel_err(s[idx].cpu().numpy(), want).max() <= tol, rel_err(s[idx].cpu().numpy(), want).max()


def test_deferred_status_raises_at_exit():
    """engine.deferred_status: the per-call status read-back is skipped inside the context, the accumulated bits are raised when it exits."""
    c = load_case("knrm", KNRM_CASES[0])
    r = KNRM({"gradkernels": True, "scoretanh": False, "singlefc": True, "finetune": False}, SimpleNamespace(embeddings=c["emb"]))
    r.build_model().to(DEV).eval()
    b = _batch(c)
    bad = dict(b, posdoc=b["posdoc"].clone())
    bad["posdoc"][1, 3] = int(c["V"]) + 5
    with torch.no_grad():
        with pytest.raises(IndexError):
            with engine.deferred_status(DEV):
                r.test(bad)           # no exception here
                s = r.test(b)         # ... and later calls still run
                assert torch.isfinite(s).all()
        assert torch.isfinite(r.test(b)).all()   # the word was cleared when the error was raised


@pytest.mark.parametrize("kind", ["drmmtks", "pacrr", "convknrm"])
def test_resident_store_serves_the_sibling_models(kind):
    """Row N1 for the row-N4 models: int32 tables + index pairs (id rows gathered on the device) give bit-identical scores to the
    int64 [B, Q] / [B, L] layout, in any pair order, and `PytorchTrainer.predict_resident` returns them rounded to fp16."""
    from capreolus_amd.feeder import CandidateStore
    from capreolus_amd.reranker import DRMMTKS
    from capreolus_amd.trainer import PytorchTrainer

    c = load_case(kind, "ranklist")
    if kind == "drmmtks":
        r = DRMMTKS({"topk": int(c["topk"])}, SimpleNamespace(embeddings=c["emb"]))
        m = r.build_model()
        m.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in c.items() if k.startswith("sd.")}, strict=False)
        m.to(DEV).eval()
    else:
        r = _pacrr_reranker(c) if kind == "pacrr" else _convknrm_reranker(c)
    B = c["query"].shape[0]
    store = CandidateStore(DEV)
    for i in range(B):   # (the fixture draws an idf row per pair, so every pair gets its own query row)
        store.add_query(f"q{i}", c["query"][i], c["query_idf"][i])
        store.add_doc(f"d{i}", c["posdoc"][i])
    store.finalize()
    pq = torch.arange(B, dtype=torch.int32, device=DEV)
    pd = torch.as_tensor([store.drow[f"d{i}"] for i in range(B)], dtype=torch.int32, device=DEV)
    with torch.no_grad():
        want = r.test(_batch(c))
        assert torch.equal(r.test_resident(store, pq, pd), want)
        perm = torch.randperm(B, device=DEV)
        assert torch.equal(r.test_resident(store, pq[perm], pd[perm]), want[perm])
    preds = PytorchTrainer({"evalbatch": 64}).predict_resident(r, store, {f"q{i}": [f"d{i}"] for i in range(B)})
    got = np.array([preds[f"q{i}"][f"d{i}"] for i in range(B)], dtype=np.float16)
    assert (got == want.cpu().numpy().astype(np.float16)).all()
    assert (got == c["ref_scores_f16"]).mean() > 0.98


def test_drmmtks_training_step_matches_autograd():
    """Row N3 for DRMM-TKS: the [B, Q, topk] features of the HIP kernel are the top-k rows of the oracle's similarity matrix (bit
    exact), and the loss / gradients of the small net on top equal the same ATen ops under autograd on the host."""
    from capreolus_amd.reranker import DRMMTKS

    c = load_case("drmmtks", "default")
    K = int(c["topk"])
    r = DRMMTKS({"topk": K}, SimpleNamespace(embeddings=c["emb"]))
    m = r.build_model()
    m.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in c.items() if k.startswith("sd.")}, strict=False)
    m.to(DEV).train()
    b = _batch(c)
    negdoc = c["posdoc"][np.roll(np.arange(len(c["posdoc"])), 1)]
    s = r.score({**b, "negdoc": _t(negdoc)})
    loss = torch.clamp(1.0 - (s[0] - s[1]), min=0).mean() + 0.01 * s[0].sum()
    loss.backward()
    packed = oracle.pack(c["emb"])
    feats = engine.drmmtks_features(b["query"], b["posdoc"], m._packed.get(m.embedding.weight), int(c["V"]), int(c["D"]), K).cpu().numpy()
    sim, err = oracle.simmat(c["query"], c["posdoc"], packed, int(c["D"]))
    assert err == 0
    assert np.array_equal(feats, -np.sort(-sim, axis=-1)[:, :, :K])                        # torch.topk values: sorted descending
    t = {k[3:]: torch.as_tensor(v).clone().requires_grad_(True) for k, v in c.items() if k.startswith("sd.")}

    def host_scores(doc):
        sm, e = oracle.simmat(c["query"], doc, packed, int(c["D"]))
        assert e == 0
        top = torch.as_tensor(-np.sort(-sm, axis=-1)[:, :, :K].copy())
        z = torch.tanh(top @ t["ffw.0.weight"].t() + t["ffw.0.bias"]).squeeze(-1)           # DRMMTKS.py:57
        q = torch.as_tensor(c["query"])
        gl = torch.as_tensor(c["query_idf"]) * t["gates.weight"].view(-1)[0] + (q == 0).float() * -1e7   # :38-41
        return (torch.softmax(gl, dim=1) * z).sum(1) * t["output_layer.weight"].view(-1)[0] + t["output_layer.bias"].view(-1)[0]

    ps, ns = host_scores(c["posdoc"]), host_scores(negdoc)
    ref = torch.clamp(1.0 - (ps - ns), min=0).mean() + 0.01 * ps.sum()
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-4 * max(1.0, abs(ref.item()))
    for name, mod in (("ffw.0.weight", m.ffw[0].weight), ("gates.weight", m.gates.weight), ("output_layer.weight", m.output_layer.weight),
                      ("ffw.0.bias", m.ffw[0].bias)):
        g, gr = mod.grad.cpu(), t[name].grad
        assert (g - gr).abs().max() <= 2e-3 * (float(gr.abs().max()) + 1e-8), name


def test_pacrr_training_step_matches_autograd():
    """Row N3 for PACRR: in training mode the similarity matrix comes from the HIP kernel and the convolutions / k-max / combine run
    under autograd; in eval mode everything is the fused kernel.  The two forwards agree, and the loss and gradients equal the same ATen
    ops applied to the oracle's similarity matrix on the host."""
    import torch.nn.functional as F

    c = load_case("pacrr", "default")
    r = _pacrr_reranker(c)
    m = r.model
    b = _batch(c)
    with torch.no_grad():
        ev = r.test(b)
    m.train()
    negdoc = c["posdoc"][np.roll(np.arange(len(c["posdoc"])), 1)]
    s = r.score({**b, "negdoc": _t(negdoc)})
    assert (s[0].detach() - ev).abs().max() <= 2e-5 * ev.abs().max()          # same scores as the fused inference kernel
    loss = torch.clamp(1.0 - (s[0] - s[1]), min=0).mean() + 0.01 * s[0].sum()
    loss.backward()
    t = {k[3:]: torch.as_tensor(v).clone().requires_grad_(True) for k, v in c.items() if k.startswith("sd.") and not k.startswith("sd.combine")}
    packed = oracle.pack(c["emb"])
    lo, hi, kmax = int(c["cfg.mingram"]), int(c["cfg.maxgram"]), int(c["cfg.kmax"])
    act = {"relu": torch.relu, "tanh": torch.tanh, "none": lambda v: v}[str(c["nonlinearity"])]

    def host_scores(doc):
        sm, e = oracle.simmat(c["query"], doc, packed, int(c["D"]))
        assert e == 0
        x = torch.as_tensor(sm).unsqueeze(1)
        feats = []
        for i, g in enumerate(range(lo, hi + 1)):
            conv = F.conv2d(F.pad(x, (0, g - 1, 0, g - 1)), t[f"ngrams.{i}.conv.weight"], t[f"ngrams.{i}.conv.bias"])
            feats.append(torch.relu(conv).max(dim=1)[0].topk(kmax, dim=2)[0])
        if bool(int(c["cfg.idf"])):
            feats.append(torch.softmax(torch.as_tensor(c["query_idf"]), dim=1).unsqueeze(2))
        h = torch.cat(feats, dim=2).reshape(x.shape[0], -1)
        h = act(h @ t["linear1.weight"].t() + t["linear1.bias"])
        h = act(h @ t["linear2.weight"].t() + t["linear2.bias"])
        return (h @ t["linear3.weight"].t() + t["linear3.bias"]).view(-1)

    ps, ns = host_scores(c["posdoc"]), host_scores(negdoc)
    ref = torch.clamp(1.0 - (ps - ns), min=0).mean() + 0.01 * ps.sum()
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-4 * max(1.0, abs(ref.item()))
    for name, par in (("ngrams.0.conv.weight", m.ngrams[0].conv.weight), ("ngrams.2.conv.weight", m.ngrams[2].conv.weight),
                      ("ngrams.1.conv.bias", m.ngrams[1].conv.bias), ("linear1.weight", m.linear1.weight), ("linear3.weight", m.linear3.weight)):
        g, gr = par.grad.cpu(), t[name].grad
        assert (g - gr).abs().max() <= 2e-3 * (float(gr.abs().max()) + 1e-8), name


# ---- training parity against the REFERENCE's own autograd (tests/golden/make_golden_grad.py) ---------------------------------------
def _tks_reranker(c):
    from capreolus_amd.reranker import DRMMTKS

    r = DRMMTKS({"topk": int(c["topk"])}, SimpleNamespace(embeddings=c["emb"]))
    m = r.build_model()
    m.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in c.items() if k.startswith("sd.")}, strict=False)
    m.to(DEV).eval()
    return r


def _tks_unfrozen_reranker(c):
    from capreolus_amd.reranker import DRMMTKS

    r = DRMMTKS({"topk": int(c["topk"]), "freezeemb": False}, SimpleNamespace(embeddings=c["emb"]))      # DRMMTKS.py:25: the table trains too
    m = r.build_model()
    m.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in c.items() if k.startswith("sd.")}, strict=False)
    m.to(DEV).eval()
    assert m.embedding.weight.requires_grad and not r.fused_step_available(32)
    return r


def _knrm_finetune_model(c):
    r = _knrm_model(c)
    r.model.embedding.weight.requires_grad = True      # finetune=True (KNRM.py:23)
    return r


REF_GRAD_CASES = [("knrm", "default", _knrm_model), ("knrm", "twolayer_tanh", _knrm_model), ("knrm", "finetune_glove50_short", _knrm_finetune_model),
                  ("drmmtks", "unfrozen_top3_short", _tks_unfrozen_reranker), ("drmm", "zero_idf", _drmm_model),
                  ("drmmtks", "default", _tks_reranker), ("pacrr", "default", _pacrr_reranker), ("pacrr", "tanh_noidf_short", _pacrr_reranker),
                  ("convknrm", "default", _convknrm_reranker), ("convknrm", "nocross_2fc_short", _convknrm_reranker)]


@pytest.mark.parametrize("kind,name,build", REF_GRAD_CASES, ids=[f"{k}-{n}" for k, n, _ in REF_GRAD_CASES])
def test_training_gradients_match_the_reference(kind, name, build):
    """Row N3 against the reference itself: `.grad` of every trainable parameter after the reference trainer's pairwise hinge loss
    (reranker/common.py:101-103; + 0.01 x the sum of the positive scores) on (documents, the same documents rolled by one pair),
    computed by the REFERENCE nn.Module under autograd on the host (fixtures `<model>_grad_<case>.npz`), against `Reranker.score()`
    in train mode on the GPU.  Loss to 1e-4, every gradient to 2e-3 of its own largest entry."""
    import os

    from tests.helpers import GOLDEN

    trains_table = name.startswith(("finetune_", "unfrozen_"))             # (the same inputs and weights as the plain case, the table trainable)
    c = load_case(kind, name.split("_", 1)[1] if trains_table else name)
    g = np.load(os.path.join(GOLDEN, f"{kind}_grad_{name}.npz"))
    r = build(c)
    m = r.model
    m.train()
    b = _batch(c)
    pos, neg = r.score({**b, "negdoc": torch.roll(b["posdoc"], 1, 0)})
    assert rel_err(pos.detach().cpu().numpy(), g["ref_pos_scores"]).max() <= REL_TOL
    assert rel_err(neg.detach().cpu().numpy(), g["ref_neg_scores"]).max() <= REL_TOL
    loss = torch.nn.functional.margin_ranking_loss(pos, neg, torch.ones_like(pos), margin=1.0) + 0.01 * pos.sum()
    loss.backward()
    assert abs(loss.item() - float(g["ref_loss"])) <= 1e-4 * max(1.0, abs(float(g["ref_loss"])))
    params = dict(m.named_parameters())
    checked = 0
    for key in g.files:
        if not key.startswith("ref_grad."):
            continue
        want = g[key]
        p = params[key[9:]]
        assert p.grad is not None, key
        got = p.grad.detach().cpu().numpy().reshape(want.shape)
        scale = float(np.abs(want).max())
        if key.startswith("ref_grad.kernels.kernels.") and float(c["sd." + key[9:].rsplit(".", 1)[0] + ".sigma"]) < 0.01:
            # the exact-match kernel (mu = 1, sigma = 0.001): dK/dmu = K (s - mu) / sigma^2 with s - mu = a few ulp of cos(a, a) - the
            # reference's own value is 1e6 x its rounding noise (the DRMM coin flip in another guise), so only its magnitude is checked
            assert float(np.abs(got).max()) <= max(20.0 * scale, 5e-3), (key, float(np.abs(got).max()), scale)
            checked += 1
            continue
        if scale == 0.0:
            assert float(np.abs(got).max()) <= 1e-7, key
        else:
            # (the finetune route is the reference's own op sequence on the GPU: what differs is the order of fp32 reductions, and a kernel
            # gradient that is a difference of nearly equal document sums shows it at 2.2e-3)
            tol = 5e-3 if trains_table else 2e-3
            assert float(np.abs(got - want).max()) <= tol * scale, (key, float(np.abs(got - want).max()), scale)
        checked += 1
    assert checked >= 5


def test_predict_builds_its_candidate_store_on_first_use():
    """Row N1 through the reference's own call site: `PytorchTrainer.predict(reranker, sampler)` - unchanged for the caller
    (task/rerank.py:108-124) - tokenises a PredSampler-contract dataset once, uploads the distinct id rows and scores by index pairs;
    later calls on the same sampler never touch the host per sample.  Predictions are the DataLoader route's, bit for bit."""
    from capreolus_amd.trainer import PytorchTrainer

    c = load_case("knrm", "ranklist")
    r = _knrm_model(c)
    B = c["query"].shape[0]
    # three queries; a document shared by two of them; ragged lists
    q2d = {"7": [f"d{i}" for i in range(0, 90)], "3": [f"d{i}" for i in range(80, 150)], "12": [f"d{i}" for i in range(150, B)]}
    qrow = {"7": 0, "3": 1, "12": 2}
    walked = [0]

    class Sampler(torch.utils.data.IterableDataset):
        qid_to_docids = q2d

        def __iter__(self):
            for qid, docs in q2d.items():
                for d in docs:
                    walked[0] += 1
                    i = int(d[1:])
                    yield {"qid": qid, "posdocid": d, "query": c["query"][qrow[qid]], "posdoc": c["posdoc"][i],
                           "query_idf": c["query_idf"][qrow[qid]]}

        def __len__(self):
            return sum(len(v) for v in q2d.values())

        def get_qid_docid_pairs(self):
            return ((q, d) for q, docs in q2d.items() for d in docs)

    s = Sampler()
    n = len(s)
    reference_route = PytorchTrainer({"batch": 32, "resident": False, "coalesce": 0}).predict(r, s)
    t = PytorchTrainer({"batch": 32})
    walked[0] = 0
    first = t.predict(r, s)
    assert walked[0] == n and first == reference_route
    second = t.predict(r, s)
    assert walked[0] == n and second == reference_route          # no second walk over the samples
    plan = next(iter(t._resident_plans.values()))[2]
    assert plan[0].d_table.shape[0] == B - 0 and plan[0].q_table.shape[0] == 3     # shared documents d80..d89 stored once
    # the candidate lists change in place: the plan is rebuilt, not reused
    q2d["12"] = q2d["12"][:-5]
    third = t.predict(r, s)
    assert walked[0] == 2 * n - 5 and third == PytorchTrainer({"batch": 32, "resident": False}).predict(r, s)
    q2d["12"] = [f"d{i}" for i in range(150, B)]
    # ... also when the edit keeps the list object, its length and every docid the sampled fingerprint looks at (the default
    # `resident_verify` = "auto" compares every docid of a run of this size with the plan's own copy)
    fourth = t.predict(r, s)
    lst = q2d["12"]
    step = max(1, (len(lst) - 1) // 7)
    assert 2 % step and 3 % step and len(lst) > 5     # positions the 8-docid sample skips
    lst[2], lst[3] = "d10", lst[2]                    # one candidate replaced by another document, one moved
    fifth = t.predict(r, s)
    assert fifth == PytorchTrainer({"batch": 32, "resident": False}).predict(r, s)
    assert "d10" in fifth["12"] and "d10" not in fourth["12"] and list(fifth["12"])[:4] == lst[:4]
    q2d["12"] = [f"d{i}" for i in range(150, B)]

    # a sampler whose document rows depend on the query they come with is not a candidate-store sampler: DataLoader route, same answers
    class Coupled(Sampler):
        def __iter__(self):
            for k, sample in enumerate(Sampler.__iter__(self)):
                if sample["posdocid"] == "d85" and sample["qid"] == "3":
                    sample = {**sample, "posdoc": np.roll(sample["posdoc"], 1)}
                yield sample

    cs = Coupled()
    assert PytorchTrainer({"batch": 32}).predict(r, cs) == PytorchTrainer({"batch": 32, "resident": False}).predict(r, cs)


@pytest.mark.parametrize("kind,name,build", [("knrm", "twolayer_tanh", _knrm_model), ("drmm", "zero_idf", _drmm_model), ("convknrm", "nocross_2fc_short", _convknrm_reranker)],
                         ids=["knrm", "drmm", "convknrm"])
def test_graphed_training_steps_equal_eager_steps(kind, name, build):
    """Row N3: `PytorchTrainer.single_train_iteration` replays ONE captured HIP graph per batch (score on positives and negatives, loss,
    backward, Adam).  Five steps through the graph leave the parameters where five eager steps leave them (the capture's warm-up steps
    are undone), with the per-step learning-rate schedule applied through the device scalar the captured Adam reads."""
    import contextlib

    from capreolus_amd.trainer import PytorchTrainer

    c = load_case(kind, name)
    B = c["query"].shape[0]
    rs = np.random.RandomState(5)
    batches = []
    for _ in range(5):
        perm = rs.permutation(B)
        batches.append({"qid": [str(i) for i in range(B)], "query": torch.as_tensor(c["query"]), "query_idf": torch.as_tensor(c["query_idf"]),
                        "posdoc": torch.as_tensor(c["posdoc"]), "negdoc": torch.as_tensor(c["posdoc"][perm])})

    def run(graph):
        r = build(c)
        m = r.model
        m.train()
        t = PytorchTrainer({"batch": B, "itersize": 5 * B, "lr": 0.01, "warmupiters": 1, "decay": 0.5, "decaytype": "linear", "graph": graph})
        t.device, t.scaler, t._train_autocast, t.loss = torch.device(DEV), None, contextlib.nullcontext, t.pair_hinge_loss
        t._train_graph, t._graph_failed = None, False
        params = [p for p in m.parameters() if p.requires_grad]
        # (the same Adam implementation on both routes: with capturable = False the bias corrections are float64 host scalars, and a
        # component whose gradient is rounding noise moves by +-lr whichever way that noise falls - not what this test is about)
        t.optimizer = torch.optim.Adam(params, lr=torch.tensor(0.01, device=DEV), capturable=True)
        t._set_lr(0)
        loss = t.single_train_iteration(r, batches, cur_iter=1)
        assert (t._train_graph is not None) == graph
        return float(loss), {k: v.detach().cpu().clone() for k, v in m.named_parameters() if v.requires_grad}

    loss_e, eager = run(False)
    loss_g, graphed = run(True)
    # (the iteration's mean loss: ConvKNRM's follows its parameters, which MIOpen's backward moves by rounding noise from run to run - seen
    # once in twelve runs of the whole suite: 1.2e-6 apart at a loss of 0.95 - so it gets the bound its parameters get below)
    assert abs(loss_e - loss_g) <= (5e-5 if kind == "convknrm" else 1e-6) * max(1.0, abs(loss_e)), (loss_e, loss_g)
    # (same kernels on both routes; MIOpen's convolution backward and ATen's reductions are not bit-reproducible run to run, and Adam turns a
    # gradient component that is pure rounding noise into a step of +-lr whichever way the noise falls: ConvKNRM, whose convolutions run in
    # MIOpen, gets the wider bound - and, since one run in fifteen of the whole suite saw a component beyond it, the bound is calibrated on the
    # spot: a SECOND eager run says how far eager is from eager on this box today, and the graph route is allowed four times that)
    # ... and a component that did flip is told from a wrong step by its size and its company: at most 2 lr per flipped step (0.02 here), in at
    # most one element in a thousand of the tensor - a graph that replayed a stale batch or skipped a step moves whole tensors
    eager2 = run(False)[1] if kind == "convknrm" else eager
    moved = 0.0
    for k, v in eager.items():
        scale = float(v.abs().max()) + 1e-6
        tol = 5e-3 if kind == "convknrm" else 2e-4
        noise = float((eager2[k] - v).abs().max())
        diff = (graphed[k] - v).abs()
        bound = max(tol * scale, 4 * noise)
        if kind == "convknrm":
            beyond = int((diff > bound).sum())
            assert beyond <= max(1, v.numel() // 1000) and float(diff.max()) <= max(bound, 0.05), (k, beyond, float(diff.max()), scale, noise)
        else:
            assert float(diff.max()) <= bound, (k, float(diff.max()), scale, noise)
        moved = max(moved, float((v - torch.as_tensor(np.asarray(c["sd." + k])).reshape(v.shape)).abs().max()) if ("sd." + k) in c else 1.0)
    assert moved > 1e-3          # the five steps did train something


@pytest.mark.parametrize("name,softmax", [("default", False), ("ranklist", True), ("glove50_short", False)])
def test_knrm_fused_training_steps_equal_eager_steps(name, softmax):
    """Row N3 as one device step (capamd_knrm_train_step: score(pos), score(neg), the pairwise loss, backward through `combine` and the RBF
    kernels, Adam - two launches, no autograd): five steps leave every parameter AND the optimizer's state where five eager steps of the
    reference's own sequence leave them - reranker.score() under autograd, the trainer's loss, loss.backward(), torch.optim.Adam.step()
    (plain Adam on both sides, with the per-step learning-rate schedule)."""
    import contextlib

    from capreolus_amd.trainer import PytorchTrainer

    c = load_case("knrm", name)
    B = min(32, c["query"].shape[0])
    rs = np.random.RandomState(5)
    batches = []
    for _ in range(5):
        perm = rs.permutation(c["query"].shape[0])
        batches.append({"qid": [str(i) for i in range(B)], "query": torch.as_tensor(c["query"][:B]), "query_idf": torch.as_tensor(c["query_idf"][:B]),
                        "posdoc": torch.as_tensor(c["posdoc"][:B]), "negdoc": torch.as_tensor(c["posdoc"][perm[:B]])})

    def run(fused):
        r = _knrm_model(c)
        if not r.model.p["singlefc"]:
            pytest.skip("two-layer combine keeps the autograd route")
        m = r.model
        m.train()
        t = PytorchTrainer({"batch": B, "itersize": 5 * B, "lr": 0.01, "warmupiters": 1, "decay": 0.5, "decaytype": "linear", "graph": False, "fused": fused,
                            "softmaxloss": softmax})
        t.device, t.scaler, t._train_autocast = torch.device(DEV), None, contextlib.nullcontext
        t.loss = t.pair_softmax_loss if softmax else t.pair_hinge_loss
        t._train_graph, t._graph_failed, t._fused_failed = None, False, False
        t._use_fused = t._fused_allowed(r)
        assert t._use_fused == fused
        t.optimizer = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=0.01)
        t._set_lr(0)
        loss = t.single_train_iteration(r, batches, cur_iter=1)
        assert not t._fused_failed
        sd = t.optimizer.state_dict()
        return float(loss), {k: v.detach().cpu().clone() for k, v in m.named_parameters() if v.requires_grad}, sd

    loss_e, eager, sd_e = run(False)
    loss_f, fused, sd_f = run(True)
    assert abs(loss_e - loss_f) <= 2e-6 * max(1.0, abs(loss_e)), (loss_e, loss_f)
    moved = 0.0
    # The Linear's bias is added to the positive AND the negative score: without the final tanh its gradient under a pairwise loss is
    # exactly zero.  The fused step computes that zero; autograd sums +1/B and -1/B over the batch's rows, and Adam turns the rounding
    # residue of that sum (1e-9) into a step of +-lr whichever way it falls - the reference's bias "trains" on noise.  Not compared.
    noise = {"combine.0.bias"} if not bool(c["scoretanh"]) else set()
    for k, v in eager.items():
        init = torch.as_tensor(np.asarray(c["sd." + k])).reshape(v.shape)
        if k in noise:
            assert float((fused[k] - init).abs().max()) == 0.0 and float((v - init).abs().max()) <= 5 * 0.01 * 1.001
            continue
        scale = float(v.abs().max()) + 1e-6
        # (the exact-match kernel, sigma = 0.001: d K / d mu = K (s - mu) / sigma^2 with s - mu a few ulp of cos(a, a) - per document a
        # rounding residue x 1e6 of either sign, so the batch sum cancels by orders of magnitude and depends on its order to 1e-3)
        exact = k.startswith("kernels.kernels.") and float(c["sd." + k.rsplit(".", 1)[0] + ".sigma"]) < 0.01
        # (everything else: a gradient is a sum over the batch of per-document terms of either sign - (f_neg - f_pos) / B for the Linear's
        # weights, which cancels to rounding residue for a kernel whose feature barely differs between documents - the two routes add
        # the terms in different orders, and Adam divides the result by its own running magnitude: 1e-3 of the parameter's scale, i.e.
        # 0.5 % of what five steps of lr = 0.01 can move it.  The losses of the five steps agree to 2e-6.)
        tol = 5e-3 if exact else 1e-3
        assert float((fused[k] - v).abs().max()) <= tol * scale, (k, float((fused[k] - v).abs().max()), scale)
        moved = max(moved, float((v - init).abs().max()))
        if exact:
            noise.add(k)
    assert moved > 1e-3          # the five steps did train something
    # the optimizer's state is a plain Adam state on both routes: same step counts, same moments
    assert sd_e["param_groups"][0]["lr"] == pytest.approx(sd_f["param_groups"][0]["lr"])
    names = [k for k, v in _knrm_model(c).model.named_parameters() if v.requires_grad]
    for i, st in sd_e["state"].items():
        assert float(st["step"]) == float(sd_f["state"][i]["step"]) == 5.0
        if names[i] in noise:
            continue
        for key in ("exp_avg", "exp_avg_sq"):
            a, b = st[key].cpu(), sd_f["state"][i][key].cpu()
            # (the moments are running means of the gradients: the same order-of-summation residue as the parameters above)
            assert float((a - b).abs().max()) <= 2e-3 * (float(a.abs().max()) + 1e-12) + 1e-12, (i, key)


def test_fused_step_is_chosen_only_where_the_reranker_takes_the_configuration():
    """`PytorchTrainer._fused_allowed`: KNRM's two-layer `combine` and DRMM's term-vector gate have no fused step - they keep the
    captured-graph route (and its capturable Adam) instead of falling back to a plain Adam stepping eagerly."""
    from capreolus_amd.trainer import PytorchTrainer

    t = PytorchTrainer({"batch": 32})
    t.device, t.scaler = torch.device(DEV), None
    assert t._fused_allowed(_knrm_model(load_case("knrm", "default")))
    assert not t._fused_allowed(_knrm_model(load_case("knrm", "twolayer_tanh")))
    assert t._fused_allowed(_drmm_model(load_case("drmm", "default")))
    assert not t._fused_allowed(_drmm_model(load_case("drmm", "tv_nh")))
    assert t._fused_allowed(_pacrr_reranker(load_case("pacrr", "default")))
    assert t._fused_allowed(_convknrm_reranker(load_case("convknrm", "default")))
    assert not t._fused_allowed(_convknrm_reranker(load_case("convknrm", "nocross_2fc_short")))
    big = PytorchTrainer({"batch": 256, "itersize": 512})
    big.device, big.scaler = torch.device(DEV), None
    assert big._fused_allowed(_knrm_model(load_case("knrm", "default"))) and not big._fused_allowed(_drmm_model(load_case("drmm", "default")))
    off = PytorchTrainer({"batch": 32, "fused": False})
    off.device, off.scaler = torch.device(DEV), None
    assert not off._fused_allowed(_knrm_model(load_case("knrm", "default")))


@pytest.mark.parametrize("name,softmax", [("default", False), ("ranklist", True), ("top3_short", False)])
def test_drmmtks_fused_training_steps_equal_eager_steps(name, softmax):
    """DRMM-TKS's training step as two launches (capamd_drmmtks_train_step: top-k features, Linear / tanh, idf gate, output layer, the
    pairwise loss, backward, Adam) against five eager steps of reranker.score() under autograd with the same plain Adam."""
    import contextlib

    from capreolus_amd.reranker import DRMMTKS
    from capreolus_amd.trainer import PytorchTrainer

    c = load_case("drmmtks", name)
    B = min(32, c["query"].shape[0])
    rs = np.random.RandomState(5)
    batches = []
    for _ in range(5):
        perm = rs.permutation(c["query"].shape[0])
        batches.append({"qid": [str(i) for i in range(B)], "query": torch.as_tensor(c["query"][:B]).clamp(min=0), "query_idf": torch.as_tensor(c["query_idf"][:B]),
                        "posdoc": torch.as_tensor(c["posdoc"][:B]), "negdoc": torch.as_tensor(c["posdoc"][perm[:B]])})

    def run(fused):
        r = DRMMTKS({"topk": int(c["topk"])}, SimpleNamespace(embeddings=c["emb"]))
        m = r.build_model()
        m.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in c.items() if k.startswith("sd.")}, strict=False)
        m.to(DEV).train()
        t = PytorchTrainer({"batch": B, "itersize": 5 * B, "lr": 0.01, "graph": False, "fused": fused, "softmaxloss": softmax})
        t.device, t.scaler, t._train_autocast = torch.device(DEV), None, contextlib.nullcontext
        t.loss = t.pair_softmax_loss if softmax else t.pair_hinge_loss
        t._train_graph, t._graph_failed, t._fused_failed = None, False, False
        t._use_fused = t._fused_allowed(r)
        assert t._use_fused == fused
        t.optimizer = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=0.01)
        t._set_lr(0)
        loss = t.single_train_iteration(r, batches, cur_iter=1)
        assert not t._fused_failed
        return float(loss), {k: v.detach().cpu().clone() for k, v in m.named_parameters() if v.requires_grad}

    loss_e, eager = run(False)
    loss_f, fused = run(True)
    assert abs(loss_e - loss_f) <= 2e-6 * max(1.0, abs(loss_e)), (loss_e, loss_f)
    moved = 0.0
    for k, v in eager.items():
        init = torch.as_tensor(np.asarray(c["sd." + k])).reshape(v.shape)
        if k == "output_layer.bias":      # its gradient under a pairwise loss is exactly zero: the reference trains it on rounding noise (see the KNRM test)
            assert float((fused[k] - init).abs().max()) == 0.0
            continue
        scale = float(v.abs().max()) + 1e-6
        assert float((fused[k] - v).abs().max()) <= 5e-4 * scale, (k, float((fused[k] - v).abs().max()), scale)
        moved = max(moved, float((v - init).abs().max()))
    assert moved > 1e-3


@pytest.mark.parametrize("name,softmax", [("default", False), ("ranklist", True)])
def test_convknrm_fused_training_steps_equal_eager_steps(name, softmax):
    """ConvKNRM's training step as device kernels only (capamd_convknrm_train_step: n-gram convolutions over the frozen table, kernel
    pooling, the single-Linear combine, the pairwise loss, backward, Adam on 2 K + 2 G + 2 parameter tensors - eleven launches, no autograd)
    against five eager steps of reranker.score() under autograd with the same plain Adam: the loss of the iteration, every parameter, the
    optimizer's moments."""
    import contextlib

    from capreolus_amd.trainer import PytorchTrainer

    c = load_case("convknrm", name)
    B = min(32, c["query"].shape[0])
    rs = np.random.RandomState(5)
    batches = []
    for _ in range(5):
        perm = rs.permutation(c["query"].shape[0])
        batches.append({"qid": [str(i) for i in range(B)], "query": torch.as_tensor(c["query"][:B]), "query_idf": torch.as_tensor(c["query_idf"][:B]),
                        "posdoc": torch.as_tensor(c["posdoc"][:B]), "negdoc": torch.as_tensor(c["posdoc"][perm[:B]])})

    def run(fused, steps=5):
        r = _convknrm_reranker(c)
        if not r.model.p["singlefc"]:
            pytest.skip("two-layer combine keeps the autograd route")
        m = r.model
        m.train()
        t = PytorchTrainer({"batch": B, "itersize": steps * B, "lr": 0.01, "graph": False, "fused": fused, "softmaxloss": softmax})
        t.device, t.scaler, t._train_autocast = torch.device(DEV), None, contextlib.nullcontext
        t.loss = t.pair_softmax_loss if softmax else t.pair_hinge_loss
        t._train_graph, t._graph_failed, t._fused_failed = None, False, False
        t._use_fused = t._fused_allowed(r)
        assert t._use_fused == fused
        t.optimizer = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=0.01)
        t._set_lr(0)
        loss = t.single_train_iteration(r, batches[:steps], cur_iter=1)
        assert not t._fused_failed
        with torch.no_grad():      # the scoring kernel's projection tables follow the weights the step kernels wrote: a fresh model with these weights scores alike
            b0 = {k: v.to(DEV) if torch.is_tensor(v) else v for k, v in batches[0].items()}
            after = r.test(b0).cpu()
            fresh = _convknrm_reranker(c)
            fresh.model.load_state_dict(m.state_dict())
            assert torch.equal(fresh.test(b0).cpu(), after)
        return float(loss), {k: v.detach().cpu().clone() for k, v in m.named_parameters() if v.requires_grad}, t.optimizer.state_dict(), after

    # ONE step: Adam's first moments are (1 - beta1) x the gradients - the two routes' gradients of every parameter, element by element
    names = [k for k, v in _convknrm_reranker(c).model.named_parameters() if v.requires_grad]
    loss_e, _, sd_e, _ = run(False, 1)
    loss_f, _, sd_f, _ = run(True, 1)
    assert abs(loss_e - loss_f) <= 1e-5 * max(1.0, abs(loss_e)), (loss_e, loss_f)
    residue = {}       # per parameter: the elements whose gradient is rounding residue (a difference of equal features): Adam moves them +-lr a step whichever way the residue falls
    for i, st in sd_e["state"].items():
        a, b = st["exp_avg"].cpu(), sd_f["state"][i]["exp_avg"].cpu()
        residue[names[i]] = a.abs() <= 1e-4 * float(a.abs().max())
        if names[i] == "combine.0.bias" and not bool(int(c["cfg.scoretanh"])):
            assert float(b.abs().max()) == 0.0 and float(a.abs().max()) <= 1e-7      # exactly zero under a pairwise loss; autograd's rounding residue
            continue
        sig = names[i].rsplit(".", 1)[0] + ".sigma"
        exact = names[i].startswith("kernels.kernels.") and float(dict(_convknrm_reranker(c).model.named_parameters())[sig].detach()) < 0.01
        assert float((a - b).abs().max()) <= (2e-2 if exact else 1e-3) * float(a.abs().max()) + 1e-12, (names[i], float((a - b).abs().max()), float(a.abs().max()))
    # TWO steps: Adam's arithmetic on the second step (bias corrections at t = 2, moments carried over), while the routes' trajectories are
    # still together (the elements Adam moved on rounding residue in step 1 - +-lr whichever way the residue fell - differ by design and
    # seed a drift that grows ~20 x per step at lr = 0.01: 1e-7 after one step, 1e-6 after two, 2e-4 after five;
    # scripts/dbg/convknrm_determinism.py - each route by itself is bit-reproducible run to run)
    start = {k: v.detach().cpu().clone() for k, v in _convknrm_reranker(c).model.named_parameters()}
    loss_e, eager, sd_e, _ = run(False, 2)
    loss_f, fused, sd_f, _ = run(True, 2)
    assert abs(loss_e - loss_f) <= 1e-4 * max(1.0, abs(loss_e)), (loss_e, loss_f)
    moved = 0.0
    for k, v in eager.items():
        if k == "combine.0.bias" and not bool(int(c["cfg.scoretanh"])):      # gradient exactly zero under a pairwise loss: see the KNRM test
            assert float((fused[k] - start[k]).abs().max()) == 0.0
            continue
        scale = float(v.abs().max()) + 1e-6
        diff = (fused[k] - v).abs()
        assert float(diff.max()) <= 2 * 0.01 * 2.001
        exact = k.startswith("kernels.kernels.") and float(start[k.rsplit(".", 1)[0] + ".sigma"]) < 0.01
        if exact:
            continue          # rounding noise x 1e6 on either route
        diff = diff * (~residue[k])
        assert float((diff > 1e-3 * scale).float().mean()) <= 2e-3, (k, float((diff > 1e-3 * scale).float().mean()), float(diff.max()), scale)
        moved = max(moved, float((v - start[k]).abs().max()))
    assert moved > 1e-3          # the steps did train something
    for i, st in sd_e["state"].items():
        assert float(st["step"]) == float(sd_f["state"][i]["step"]) == 2.0
    # FIVE steps: a sanity check of where the trajectories end (see above: by then they have drifted)
    loss_e, eager, sd_e, after_e = run(False)
    loss_f, fused, sd_f, after_f = run(True)
    assert abs(loss_e - loss_f) <= 1e-2 * max(1.0, abs(loss_e)), (loss_e, loss_f)
    for k, v in eager.items():
        assert float((fused[k] - v).abs().max()) <= 5 * 0.01 * 2.001
        if k.startswith("convs."):
            assert float((fused[k] - v).abs().mean()) <= 0.05 * 5 * 0.01, (k, float((fused[k] - v).abs().mean()))
    for i, st in sd_e["state"].items():
        assert float(st["step"]) == float(sd_f["state"][i]["step"]) == 5.0
    assert torch.isfinite(after_e).all() and torch.isfinite(after_f).all()


@pytest.mark.parametrize("name,softmax", [("default", False), ("tanh_noidf_short", True), ("ranklist", False)])
def test_pacrr_fused_training_steps_equal_eager_steps(name, softmax):
    """PACRR's training step as device kernels only (capamd_pacrr_train_step: similarity matrices, the Conv2d / ReLU / max / k-max stage,
    the idf softmax, three Linear layers with their nonlinearity, the pairwise loss, backward, Adam on 2 n + 6 parameter tensors - six
    launches, no autograd) against eager steps of reranker.score() under autograd with the same plain Adam: after ONE step the optimizer's
    first moments - (1 - beta1) x the gradients - element by element, after five the parameters."""
    import contextlib

    from capreolus_amd.trainer import PytorchTrainer

    c = load_case("pacrr", name)
    B = min(32, c["query"].shape[0])
    rs = np.random.RandomState(5)
    batches = []
    for _ in range(5):
        perm = rs.permutation(c["query"].shape[0])
        batches.append({"qid": [str(i) for i in range(B)], "query": torch.as_tensor(c["query"][:B]), "query_idf": torch.as_tensor(c["query_idf"][:B]),
                        "posdoc": torch.as_tensor(c["posdoc"][:B]), "negdoc": torch.as_tensor(c["posdoc"][perm[:B]])})

    def run(fused, steps):
        r = _pacrr_reranker(c)
        m = r.model
        m.train()
        t = PytorchTrainer({"batch": B, "itersize": steps * B, "lr": 0.01, "graph": False, "fused": fused, "softmaxloss": softmax})
        t.device, t.scaler, t._train_autocast = torch.device(DEV), None, contextlib.nullcontext
        t.loss = t.pair_softmax_loss if softmax else t.pair_hinge_loss
        t._train_graph, t._graph_failed, t._fused_failed = None, False, False
        t._use_fused = t._fused_allowed(r)
        assert t._use_fused == fused
        t.optimizer = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=0.01)
        t._set_lr(0)
        loss = t.single_train_iteration(r, batches[:steps], cur_iter=1)
        assert not t._fused_failed
        return float(loss), {k: v.detach().cpu().clone() for k, v in m.named_parameters() if v.requires_grad}, t.optimizer.state_dict()

    names = [k for k, v in _pacrr_reranker(c).model.named_parameters() if v.requires_grad]
    start = {k: v.detach().cpu().clone() for k, v in _pacrr_reranker(c).model.named_parameters()}
    loss_e, _, sd_e = run(False, 1)
    loss_f, _, sd_f = run(True, 1)
    assert abs(loss_e - loss_f) <= 1e-5 * max(1.0, abs(loss_e)), (loss_e, loss_f)
    residue = {}       # the elements whose gradient is rounding residue: Adam moves them +-lr a step whichever way the residue falls
    for i, st in sd_e["state"].items():
        a, b = st["exp_avg"].cpu(), sd_f["state"][i]["exp_avg"].cpu()
        residue[names[i]] = a.abs() <= 1e-4 * float(a.abs().max())
        if names[i] == "linear3.bias":         # added to both scores of a pair: exactly zero under a pairwise loss; autograd's rounding residue
            assert float(b.abs().max()) == 0.0 and float(a.abs().max()) <= 1e-7
            continue
        assert float((a - b).abs().max()) <= 1e-3 * float(a.abs().max()) + 1e-12, (names[i], float((a - b).abs().max()), float(a.abs().max()))
    loss_e, eager, sd_e = run(False, 5)
    loss_f, fused, sd_f = run(True, 5)
    assert abs(loss_e - loss_f) <= 1e-4 * max(1.0, abs(loss_e)), (loss_e, loss_f)
    moved = 0.0
    for k, v in eager.items():
        diff = (fused[k] - v).abs()
        assert float(diff.max()) <= 5 * 0.01 * 2.001
        if k == "linear3.bias":
            assert float((fused[k] - start[k]).abs().max()) == 0.0
            continue
        scale = float(v.abs().max()) + 1e-6
        diff = diff * (~residue[k])
        # (a k-max winner can change between the routes once the weights differ in their residue elements: a handful of elements of the
        # convolutions follow; the bulk is compared tightly)
        # (... which five steps of lr = 0.01 through two ReLU layers amplify: most elements stay within 2e-3 of the parameter's scale, the
        # mean difference within 2e-3 of it - a fraction of ONE Adam step of a run that took five)
        assert float((diff > 2e-3 * scale).float().mean()) <= 0.25, (k, float((diff > 2e-3 * scale).float().mean()), float(diff.max()))
        assert float(diff.mean()) <= 2e-3 * scale, (k, float(diff.mean()), scale)
        moved = max(moved, float((v - start[k]).abs().max()))
    assert moved > 1e-3
    for i, st in sd_e["state"].items():
        assert float(st["step"]) == float(sd_f["state"][i]["step"]) == 5.0


@pytest.mark.parametrize("kind", ["convknrm", "pacrr"])
@pytest.mark.parametrize("B", [1, 5])
def test_fused_steps_take_small_and_odd_batches(kind, B):
    """The device-kernel training steps of ConvKNRM and PACRR at batch sizes that fill no tile, chunk or wave evenly (1 pair, 5 pairs): the
    step's loss and, through Adam's first moments, every gradient against the autograd route."""
    import contextlib

    from capreolus_amd.trainer import PytorchTrainer

    c = load_case(kind, "default")
    build = _convknrm_reranker if kind == "convknrm" else _pacrr_reranker
    batch = {"qid": [str(i) for i in range(B)], "query": torch.as_tensor(c["query"][:B]), "query_idf": torch.as_tensor(c["query_idf"][:B]),
             "posdoc": torch.as_tensor(c["posdoc"][:B]), "negdoc": torch.as_tensor(c["posdoc"][B:2 * B])}

    def run(fused):
        r = build(c)
        r.model.train()
        t = PytorchTrainer({"batch": B, "itersize": B, "lr": 0.01, "graph": False, "fused": fused})
        t.device, t.scaler, t._train_autocast = torch.device(DEV), None, contextlib.nullcontext
        t.loss = t.pair_hinge_loss
        t._train_graph, t._graph_failed, t._fused_failed = None, False, False
        t._use_fused = t._fused_allowed(r)
        assert t._use_fused == fused
        t.optimizer = torch.optim.Adam([p for p in r.model.parameters() if p.requires_grad], lr=0.01)
        t._set_lr(0)
        loss = t.single_train_iteration(r, [batch], cur_iter=1)
        assert not t._fused_failed
        return float(loss), t.optimizer.state_dict(), [k for k, v in r.model.named_parameters() if v.requires_grad]

    loss_e, sd_e, names = run(False)
    loss_f, sd_f, _ = run(True)
    assert abs(loss_e - loss_f) <= 1e-5 * max(1.0, abs(loss_e)), (loss_e, loss_f)
    for i, st in sd_e["state"].items():
        a, b = st["exp_avg"].cpu(), sd_f["state"][i]["exp_avg"].cpu()
        if names[i] in ("combine.0.bias", "linear3.bias"):        # exactly zero under a pairwise loss (see the five-step tests)
            continue
        if names[i].startswith("kernels.kernels.10."):            # the exact-match kernel: rounding noise x 1e6 on either route
            continue
        assert float((a - b).abs().max()) <= 2e-3 * float(a.abs().max()) + 1e-10, (names[i], float((a - b).abs().max()), float(a.abs().max()))


@pytest.mark.parametrize("name,softmax", [("default", False), ("zero_idf", True), ("ch", False)])
def test_drmm_fused_training_steps_equal_eager_steps(name, softmax):
    """DRMM's training step as two launches (capamd_drmm_train_step: matching histograms, the 30 -> 5 -> 1 tanh net, idf gate, output layer,
    the pairwise loss, backward, Adam) against five eager steps of reranker.score() under autograd with the same plain Adam."""
    import contextlib

    from capreolus_amd.trainer import PytorchTrainer

    c = load_case("drmm", name)
    B = min(32, c["query"].shape[0])
    rs = np.random.RandomState(5)
    batches = []
    for _ in range(5):
        perm = rs.permutation(c["query"].shape[0])
        batches.append({"qid": [str(i) for i in range(B)], "query": torch.as_tensor(c["query"][:B]).clamp(min=0), "query_idf": torch.as_tensor(c["query_idf"][:B]),
                        "posdoc": torch.as_tensor(c["posdoc"][:B]), "negdoc": torch.as_tensor(c["posdoc"][perm[:B]])})

    def run(fused):
        r = _drmm_model(c)
        m = r.model
        m.train()
        t = PytorchTrainer({"batch": B, "itersize": 5 * B, "lr": 0.01, "graph": False, "fused": fused, "softmaxloss": softmax})
        t.device, t.scaler, t._train_autocast = torch.device(DEV), None, contextlib.nullcontext
        t.loss = t.pair_softmax_loss if softmax else t.pair_hinge_loss
        t._train_graph, t._graph_failed, t._fused_failed = None, False, False
        t._use_fused = t._fused_allowed(r)
        assert t._use_fused == fused
        t.optimizer = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=0.01)
        t._set_lr(0)
        loss = t.single_train_iteration(r, batches, cur_iter=1)
        assert not t._fused_failed
        return float(loss), {k: v.detach().cpu().clone() for k, v in m.named_parameters() if v.requires_grad}

    loss_e, eager = run(False)
    loss_f, fused = run(True)
    assert abs(loss_e - loss_f) <= 2e-6 * max(1.0, abs(loss_e)), (loss_e, loss_f)
    moved = 0.0
    for k, v in eager.items():
        init = torch.as_tensor(np.asarray(c["sd." + k])).reshape(v.shape)
        if k == "output_layer.bias":      # its gradient under a pairwise loss is exactly zero: the reference trains it on rounding noise (see the KNRM test)
            assert float((fused[k] - init).abs().max()) == 0.0
            continue
        scale = float(v.abs().max()) + 1e-6
        assert float((fused[k] - v).abs().max()) <= 5e-4 * scale, (k, float((fused[k] - v).abs().max()), scale)
        moved = max(moved, float((v - init).abs().max()))
    assert moved > 1e-3


def _graph_trainer(r, B, n_batches, graph):
    import contextlib

    from capreolus_amd.trainer import PytorchTrainer

    t = PytorchTrainer({"batch": B, "itersize": n_batches * B, "lr": 0.01, "graph": graph})
    t.device, t.scaler, t._train_autocast, t.loss = torch.device(DEV), None, contextlib.nullcontext, t.pair_hinge_loss
    t._train_graph, t._graph_failed = None, False
    t.optimizer = torch.optim.Adam([p for p in r.model.parameters() if p.requires_grad], lr=torch.tensor(0.01, device=DEV), capturable=True)
    t._set_lr(0)
    return t


def test_predict_after_graphed_iterations_uses_the_trained_weights():
    """A graph replay updates the parameters without moving `tensor._version`, which is what ConvKNRM's folded projection tables (and
    every other weight-derived cache of the engine) are keyed on: after graphed iterations `test()` must score with the CURRENT
    convolutions - equal to a freshly built model loaded with the same weights - not with tables folded before the training."""
    c = load_case("convknrm", "nocross_2fc_short")
    B = c["query"].shape[0]
    rs = np.random.RandomState(9)
    batch = lambda: {"qid": [str(i) for i in range(B)], "query": torch.as_tensor(c["query"]), "query_idf": torch.as_tensor(c["query_idf"]),  # noqa: E731
                     "posdoc": torch.as_tensor(c["posdoc"]), "negdoc": torch.as_tensor(c["posdoc"][rs.permutation(B)])}
    r = _convknrm_reranker(c)
    d = _batch(c)
    r.model.eval()
    with torch.no_grad():
        before = r.test(d).clone()           # folds the projection tables from the initial weights
    t = _graph_trainer(r, B, 3, True)
    for it in (1, 2):
        r.model.train()
        t.single_train_iteration(r, [batch() for _ in range(3)], cur_iter=it)
        assert t._train_graph is not None
        r.model.eval()
        with torch.no_grad():
            got = r.test(d).clone()
        fresh = _convknrm_reranker(c)
        fresh.model.load_state_dict({k: v for k, v in r.model.state_dict().items() if "embedding" not in k}, strict=False)
        fresh.model.eval()
        with torch.no_grad():
            want = fresh.test(d)
        assert torch.equal(got, want), float((got - want).abs().max())
        assert float((got - before).abs().max()) > 1e-4          # ... and the training did move the scores


@pytest.mark.parametrize("kind,name,build", [("knrm", "twolayer_tanh", _knrm_model), ("drmm", "zero_idf", _drmm_model)], ids=["knrm", "drmm"])
def test_short_batch_after_graph_replays_takes_a_clean_eager_step(kind, name, build):
    """A batch of another shape (the short last batch) runs eagerly between replays: its backward must not add to the graph's static
    gradient tensors, which still hold the previous replay's gradients - full, full, short, full equals the same four eager steps."""
    c = load_case(kind, name)
    B = c["query"].shape[0]
    rs = np.random.RandomState(6)

    def mk(n):
        perm = rs.permutation(B)[:n]
        return {"qid": [str(i) for i in range(n)], "query": torch.as_tensor(c["query"][:n]), "query_idf": torch.as_tensor(c["query_idf"][:n]),
                "posdoc": torch.as_tensor(c["posdoc"][:n]), "negdoc": torch.as_tensor(c["posdoc"][perm])}

    batches = [mk(B), mk(B), mk(max(2, B // 2)), mk(B)]
    out = {}
    for graph in (False, True):
        r = build(c)
        r.model.train()
        t = _graph_trainer(r, B, len(batches), graph)
        t.single_train_iteration(r, batches, cur_iter=1)
        assert (t._train_graph is not None) == graph
        out[graph] = {k: v.detach().cpu().clone() for k, v in r.model.named_parameters() if v.requires_grad}
    for k, v in out[False].items():
        scale = float(v.abs().max()) + 1e-6
        assert float((out[True][k] - v).abs().max()) <= 2e-4 * scale, (k, float((out[True][k] - v).abs().max()), scale)


def test_optimizer_checkpoint_is_the_plain_kind_and_resumes_a_graphed_run(tmp_path):
    """`save_weights` writes the optimizer state as the reference's plain Adam would (float lr, capturable off, host step counters) even
    when the run trains through the captured step; `load_weights` puts it back into whichever Adam the caller runs."""
    c = load_case("knrm", "twolayer_tanh")
    r = _knrm_model(c)
    params = [p for p in r.model.parameters() if p.requires_grad]
    cap = torch.optim.Adam(params, lr=torch.tensor(0.01, device=DEV), capturable=True)
    loss = sum(p.sum() for p in params)
    loss.backward()
    cap.step()
    fn = tmp_path / "w.p"
    r.save_weights(fn, cap)
    import pickle

    sd = pickle.load(open(str(fn) + ".optimizer", "rb"))
    assert all(isinstance(g["lr"], float) and not g.get("capturable") for g in sd["param_groups"])
    assert all(st["step"].device.type == "cpu" for st in sd["state"].values())
    plain = torch.optim.Adam(params, lr=0.5)
    r.load_weights(fn, plain)                       # an eager run (or the reference) resumes from it
    assert plain.param_groups[0]["lr"] == pytest.approx(0.01) and not plain.param_groups[0]["capturable"]
    cap2 = torch.optim.Adam(params, lr=torch.tensor(0.5, device=DEV), capturable=True)
    r.load_weights(fn, cap2)                        # ... and so does a graphed one: still capturable, device lr and step
    g = cap2.param_groups[0]
    assert g["capturable"] and torch.is_tensor(g["lr"]) and g["lr"].is_cuda and float(g["lr"]) == pytest.approx(0.01)
    assert all(st["step"].is_cuda and float(st["step"]) == 1.0 for st in cap2.state.values())
    for p in params:
        assert torch.equal(cap2.state[p]["exp_avg"], cap.state[p]["exp_avg"])


@pytest.mark.parametrize("N,Q,L,D,G,F", [(5, 4, 300, 300, 3, 128), (3, 3, 70, 64, 2, 32), (2, 5, 130, 48, 4, 96), (64, 4, 800, 300, 3, 128), (1, 1, 1, 8, 1, 4),
                                         (2, 4, 40, 316, 3, 160)])
def test_ngram_conv_matches_conv1d(N, Q, L, D, G, F):
    """ConvKNRM's convolution stack as the matrix-pipe kernels of ngram_conv.hip (`engine.NgramConv`) against the reference's op sequence
    (ConvKNRM.py:42-51: embeddings -> permute -> ConstantPad1d -> Conv1d -> permute) in FLOAT64 on the same tensors: the outputs at
    every real position, and the gradients of every weight and bias under an upstream gradient that is zero at pad positions (what the
    kernel pooling hands back).  Documents end in padding of different lengths, one is all padding, one has a pad inside."""
    import torch.nn.functional as Fn

    rng = np.random.default_rng(N * 1000 + L)
    V = 500
    emb = _t(rng.normal(0, 0.5, (V, D)).astype(np.float32))
    q = rng.integers(1, V, (N, Q))
    d = rng.integers(1, V, (N, L))
    for n in range(N):
        d[n, int(rng.integers(1, L + 1)):] = 0
    if N > 1:
        d[1] = 0
        q[N - 1, Q - 1] = 0
    if L > 10:
        d[0, 3] = 0
        d[0, :3] = [7, 8, 9]
    q, d = _t(q), _t(d)
    ws = [_t(rng.normal(0, 0.1, (F, D, g)).astype(np.float32)).requires_grad_() for g in range(1, G + 1)]
    bs = [_t(rng.normal(0, 0.1, (F,)).astype(np.float32)).requires_grad_() for g in range(1, G + 1)]
    wb = [t for pair in zip(ws, bs) for t in pair]
    qrep, drep = engine.NgramConv.apply(q, d, emb, *wb)
    gq = _t(rng.normal(0, 1, qrep.shape).astype(np.float32)) * (q != 0)[:, None, :, None]
    gd = _t(rng.normal(0, 1, drep.shape).astype(np.float32)) * (d != 0)[:, None, :, None]
    ((qrep * gq).sum() + (drep * gd).sum()).backward()
    got = [t.grad.clone() for t in wb]
    w64 = [w.detach().double().requires_grad_() for w in ws]
    b64 = [b.detach().double().requires_grad_() for b in bs]
    want_q, want_d = [], []
    for g in range(1, G + 1):
        for ids, out in ((q, want_q), (d, want_d)):
            x = Fn.pad(emb.double()[ids].permute(0, 2, 1), (0, g - 1))
            out.append(Fn.conv1d(x, w64[g - 1], b64[g - 1]).permute(0, 2, 1))
    want_q, want_d = torch.stack(want_q, 1), torch.stack(want_d, 1)
    ((want_q * gq.double()).sum() + (want_d * gd.double()).sum()).backward()
    for have, want, ids in ((qrep, want_q, q), (drep, want_d, d)):
        real = (ids != 0)[:, None, :, None].expand_as(have)
        assert torch.isfinite(have).all()
        err = ((have.double() - want).abs() * real).max()
        assert float(err.detach()) <= 1e-5 * float(want.detach().abs().max()), float(err.detach())
    for have, want in zip(got, [t.grad for pair in zip(w64, b64) for t in pair]):
        assert have.shape == want.shape
        assert float((have.double() - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-9, (tuple(have.shape), float((have.double() - want).abs().max()))
    bad = d.clone()
    bad[0, 0] = V
    with pytest.raises(IndexError):
        engine.NgramConv.apply(q, bad, emb, *wb)


@pytest.mark.parametrize("name", ["default", "nocross_2fc_short"])
def test_convknrm_hip_kernel_pooling_matches_autograd_through_aten(name):
    """ConvKNRM's training step behind its convolutions - cosine of every n-gram view pair, pad masks, RBF kernel pooling, log / mask /
    sum - as the HIP kernels of kernel_pool.hip (forward + backward into both convolution outputs, mu and sigma) against the reference's
    op sequence under ATen autograd (`_forward_train_aten`, itself pinned on the reference fixtures): same scores, same gradients."""
    c = load_case("convknrm", name)
    b = _batch(c)
    b["query"] = b["query"].clone()
    b["query"][1, -1] = 0          # a padded query position and an all-pad document: the masks matter
    b["posdoc"] = b["posdoc"].clone()
    b["posdoc"][2] = 0
    grads = {}
    for route in ("hip", "aten"):
        r = _convknrm_reranker(c)
        m = r.model
        m.train()
        fwd = m._forward_train if route == "hip" else m._forward_train_aten
        pos = fwd(b["posdoc"], b["query"]).view(-1)
        neg = fwd(torch.roll(b["posdoc"], 1, 0), b["query"]).view(-1)
        loss = torch.clamp(1.0 - (pos - neg), min=0).mean() + 0.01 * pos.sum()
        loss.backward()
        grads[route] = (pos.detach().cpu().numpy(), float(loss), {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None})
    (ph, lh, gh), (pa, la, ga) = grads["hip"], grads["aten"]
    assert rel_err(ph, pa).max() <= 2e-5 and abs(lh - la) <= 1e-5 * max(1.0, abs(la))
    assert set(gh) == set(ga) and any(k.startswith("convs.") for k in gh)
    for k, want in ga.items():
        scale = float(np.abs(want).max())
        if k.startswith("kernels.kernels.") and float(c["sd." + k.rsplit(".", 1)[0] + ".sigma"]) < 0.01:
            continue        # the exact-match kernel: its mu / sigma gradients are rounding noise x 1e6 on either route
        assert float(np.abs(gh[k] - want).max()) <= 1e-3 * scale + 1e-7, (k, float(np.abs(gh[k] - want).max()), scale)


@pytest.mark.parametrize("name", ["default", "tanh_noidf_short"])
def test_pacrr_hip_convmax_matches_autograd_through_aten(name):
    """PACRR's trainable stage (n-gram Conv2d -> ReLU -> max over filters -> k-max; PACRR.py:68-78) as the HIP kernels of
    pacrr_train.hip, forward with the winners' coordinates and backward into the convolution weights, against the reference's op
    sequence under ATen autograd on the same HIP similarity matrix: same scores, same gradients of every trainable parameter."""
    c = load_case("pacrr", name)
    b = _batch(c)
    out = {}
    for route in ("hip", "aten"):
        r = _pacrr_reranker(c)
        m = r.model
        m.train()
        fwd = m._forward_train if route == "hip" else m._forward_train_aten
        pos = fwd(b["posdoc"], b["query"], b["query_idf"]).view(-1)
        neg = fwd(torch.roll(b["posdoc"], 1, 0), b["query"], b["query_idf"]).view(-1)
        loss = torch.clamp(1.0 - (pos - neg), min=0).mean() + 0.01 * pos.sum()
        loss.backward()
        out[route] = (pos.detach().cpu().numpy(), float(loss), {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None})
    (ph, lh, gh), (pa, la, ga) = out["hip"], out["aten"]
    assert rel_err(ph, pa).max() <= 2e-5 and abs(lh - la) <= 1e-5 * max(1.0, abs(la))
    assert set(gh) == set(ga) and any(k.endswith("conv.weight") for k in gh)
    for k, want in ga.items():
        scale = float(np.abs(want).max())
        assert float(np.abs(gh[k] - want).max()) <= 1e-3 * scale + 1e-7, (k, float(np.abs(gh[k] - want).max()), scale)


# ---- whole candidate lists (csrc/lists.hip): every distinct term of a LIST gathered once ----------------------------------------------
def _lists_batch(n_lists, docs, V, seed, Q=4, L=800):
    rs = np.random.RandomState(seed)
    parts = [synthetic.make_candidate_list(rs, int(n), V, Q, L, same_query=True, oov_range=30, query_oov_frac=0.3) for n in docs]
    b = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    for k in ("query_idf",):       # a list's idf row is its query's
        off = 0
        for n in docs:
            b[k][off:off + n] = b[k][off]
            off += n
    return b, np.concatenate([[0], np.cumsum(docs)]).astype(np.int64)


def test_knrm_lists_match_the_per_pair_kernels():
    """capamd_knrm_forward_lists (mark -> one gather per distinct term of a list -> 16-byte lookups) against the per-pair kernel and the
    C oracle: ragged lists (1 .. 700 documents), OOV query terms with exact matches in the documents, pads, an all-pad document."""
    V, D = 3000, 300
    emb = synthetic.make_embeddings(V, D, seed=5)
    docs = [700, 1, 3, 64, 250]
    b, off = _lists_batch(len(docs), docs, V, 17)
    b["posdoc"][5] = 0
    b["posdoc"][9, 3] = b["query"][9, 0] if b["query"][9, 0] < 0 else -4     # an OOV document term (equal to the query's if that is OOV)
    r = KNRM({"singlefc": False, "scoretanh": True}, SimpleNamespace(embeddings=emb))
    torch.manual_seed(3)
    m = r.build_model().to(DEV).eval()
    d = {k: _t(v) for k, v in b.items()}
    with torch.no_grad():
        pairwise = r.test(d).cpu().numpy()
        lists = r.test_lists(d, off).cpu().numpy()
    # (the pooling sums are formed in another order: features of size ~50 move by their fp32 rounding, ~5e-6, and a random two-layer tanh
    # combine can cancel them to a score a hundred times smaller - the error is bounded against the scores' scale, as in the geometry sweep)
    scale = float(np.abs(pairwise).max())
    assert np.abs(lists - pairwise).max() <= 2e-5 * scale, (np.abs(lists - pairwise).max(), scale)
    assert np.median(rel_err(lists, pairwise)) <= 1e-6
    mu, sigma = (x.cpu().numpy() for x in m.kernels.stacked())
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want, err = oracle.knrm(b["query"], b["posdoc"], oracle.pack(emb), D, mu, sigma, sd["combine.0.weight"], sd["combine.0.bias"], sd["combine.2.weight"],
                            sd["combine.2.bias"], True)
    assert err == 0 and np.abs(lists - want).max() <= ORACLE_TOL * scale
    # the same through a candidate store (int32 tables + pair rows), lists in another order than the tables' rows
    from capreolus_amd.feeder import CandidateStore

    store = CandidateStore(DEV)
    for i in range(len(off) - 1):
        store.add_query(f"q{i}", b["query"][off[i]], b["query_idf"][off[i]])
    for j in range(b["posdoc"].shape[0]):
        store.add_doc(f"d{j}", b["posdoc"][j])
    store.finalize()
    pq = _t(np.repeat(np.arange(len(docs)), docs).astype(np.int32))
    pd = _t(np.arange(b["posdoc"].shape[0], dtype=np.int32))
    with torch.no_grad():
        via_store = r.test_resident_lists(store, pq, pd, off).cpu().numpy()
    assert np.array_equal(via_store, lists)
    # error behaviour: an id beyond the table is flagged, more than eight query terms are refused (eight are two blocks of four)
    bad = {k: v.clone() for k, v in d.items()}
    bad["posdoc"][2, 1] = V + 5
    with pytest.raises(IndexError):
        r.test_lists(bad, off)
    eight = {**d, "query": torch.cat([d["query"], d["query"]], dim=1)}
    with torch.no_grad():
        assert np.abs(r.test_lists(eight, off).cpu().numpy() - r.test(eight).cpu().numpy()).max() <= 2e-5 * scale
    with pytest.raises((EngineError, ValueError)):
        r.test_lists({**d, "query": torch.cat([d["query"]] * 3, dim=1)}, off)


@pytest.mark.parametrize("kind", ["knrm", "drmm", "drmmtks", "pacrr"])
def test_lists_of_more_than_one_query_are_refused(kind):
    """a list is scored against its FIRST pair's query (DRMM / DRMM-TKS: and that pair's idf row).  Offsets that put
    the pairs of two queries in one list - or a pair whose own query / idf row was edited - used to return the first query's scores
    silently; the mark pass now compares every pair's rows with its list's and the call raises (CAPAMD_STATUS_LIST_QUERY), by the int64
    route and through a candidate store alike.  Rows that differ only in an idf the model never reads are not an error."""
    from capreolus_amd.feeder import CandidateStore
    from capreolus_amd.reranker import DRMM

    c = load_case("knrm" if kind == "knrm" else "drmmtks" if kind in ("drmm", "drmmtks") else "pacrr", "multiquery")
    if kind == "knrm":
        r = _knrm_model(c)
    elif kind == "drmm":
        torch.manual_seed(21)
        r = DRMM({}, SimpleNamespace(embeddings=c["emb"]))
        r.build_model().to(DEV).eval()
    else:
        r = _tks_reranker(c) if kind == "drmmtks" else _pacrr_reranker(c)
    b = _batch(c)
    if kind == "drmm":
        b["query"] = b["query"].clone()
        b["query"][b["query"] < 0] = 17
    off = np.asarray(c["list_offsets"], dtype=np.int64)
    with torch.no_grad():
        good = r.test_lists(b, off)                       # the fixture's own lists: fine
        merged = np.concatenate([off[:1], off[2:]])       # lists 0 and 1 as ONE list: two queries
        assert not torch.equal(b["query"][off[0]], b["query"][off[1]])
        with pytest.raises(ValueError, match="more than one query"):
            r.test_lists(b, merged)
        assert torch.equal(r.test_lists(b, off), good)    # (the status word is clean again)
        # one pair in the middle of a list with another query term
        edited = {k: v.clone() for k, v in b.items()}
        mid = int(off[2] + 3)
        edited["query"][mid, 0] = 2 if int(edited["query"][mid, 0]) != 2 else 3
        with pytest.raises(ValueError, match="more than one query"):
            r.test_lists(edited, off)
        # ... or another idf value: an error only for the models that read the list's idf row
        edited = {k: v.clone() for k, v in b.items()}
        edited["query_idf"][mid, 0] += 0.5
        if kind in ("drmm", "drmmtks"):
            with pytest.raises(ValueError, match="more than one query"):
                r.test_lists(edited, off)
        else:
            r.test_lists(edited, off)
        # through a candidate store: the pair's query ROW index decides (equal rows under two indices are one query)
        store = CandidateStore(DEV)
        bq, bi, bd = b["query"].cpu().numpy(), b["query_idf"].cpu().numpy(), b["posdoc"].cpu().numpy()
        for i in range(len(off) - 1):
            store.add_query(f"q{i}", bq[off[i]], bi[off[i]])
        store.add_query("twin", bq[off[0]], bi[off[0]])      # the same content as q0 under another row
        for j in range(bd.shape[0]):
            store.add_doc(f"d{j}", bd[j])
        store.finalize()
        pq = np.repeat(np.arange(len(off) - 1), np.diff(off)).astype(np.int32)
        pd = _t(np.arange(bd.shape[0], dtype=np.int32))
        assert torch.equal(r.test_resident_lists(store, _t(pq), pd, off), good)
        twin = pq.copy()
        twin[int(off[0]) + 1] = len(off) - 1                  # a pair of list 0 pointing at the twin row: same query, no error
        assert torch.equal(r.test_resident_lists(store, _t(twin), pd, off), good)
        wrong = pq.copy()
        wrong[int(off[0]) + 1] = 3                            # ... at another query's row
        with pytest.raises(ValueError, match="more than one query"):
            r.test_resident_lists(store, _t(wrong), pd, off)


def test_drmm_lists_are_bit_identical_to_the_per_pair_kernels():
    """DRMM over whole lists: the similarities are computed by the same arithmetic and the bin counts are integers - scores, counts and
    fp16 rank order equal capamd_drmm_forward's bit for bit."""
    V, D = 3000, 300
    emb = synthetic.make_embeddings(V, D, seed=6)
    docs = [300, 2, 90, 1, 120]
    rs = np.random.RandomState(23)
    parts = [synthetic.make_candidate_list(rs, int(n), V, 4, 800, same_query=True, oov_range=30, query_oov_frac=0.0) for n in docs]
    b = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    off = np.concatenate([[0], np.cumsum(docs)]).astype(np.int64)
    for i in range(len(docs)):
        b["query_idf"][off[i]:off[i + 1]] = b["query_idf"][off[i]]
    b["posdoc"][4] = 0
    # (29 bins / 5 nodes: the single-wave tail of the list kernel; 40 bins or 20 nodes: its wave-per-term tail)
    for hist, gate, extra in (("LCH", "IDF", {}), ("NH", "TV", {}), ("CH", "IDF", {}), ("LCH", "TV", {"nbins": 40}), ("NH", "IDF", {"nodes": 20})):
        r = DRMM({"histType": hist, "gateType": gate, **extra}, SimpleNamespace(embeddings=emb))
        torch.manual_seed(1)
        m = r.build_model().to(DEV).eval()
        with torch.no_grad():
            m.gates.weight.mul_(30.0)
            m.ffw[0].weight.mul_(4.0)
        d = {k: _t(v) for k, v in b.items()}
        with torch.no_grad():
            pairwise = r.test(d)
            lists = r.test_lists(d, off)
        assert torch.equal(lists, pairwise), (hist, gate, extra, float((lists - pairwise).abs().max()))


@pytest.mark.parametrize("K", [5, 12])
def test_knrm_lists_with_other_kernel_banks(K):
    """The KNRM list pooling is specialised for the reference's 11-kernel bank; any other number of kernels (the C ABI takes up to 12)
    runs its general form - compared with the per-pair kernel through the engine calls, one- and two-layer read-out."""
    from capreolus_amd import engine

    V, D = 2000, 300
    emb = synthetic.make_embeddings(V, D, seed=12)
    docs = [40, 9, 130]
    b, off = _lists_batch(len(docs), docs, V, 50 + K)
    q, d = _t(b["query"]), _t(b["posdoc"])
    w = _t(emb)
    packed = engine.PackedEmbedding().get(w)
    g = torch.Generator().manual_seed(K)
    mu = _t(np.linspace(-0.9, 1.0, K).astype(np.float32))
    sigma = _t(np.r_[np.full(K - 1, 0.1), 0.001].astype(np.float32))
    for hidden in (0, 7):
        w1 = (torch.randn((hidden or 1), K, generator=g) * 0.3).to(DEV).contiguous()
        b1 = (torch.randn(hidden or 1, generator=g) * 0.1).to(DEV)
        w2 = (torch.randn(hidden, generator=g) * 0.5).to(DEV) if hidden else None
        b2 = (torch.randn(1, generator=g) * 0.1).to(DEV) if hidden else None
        a1 = w1 if hidden else w1.view(-1)
        pairwise = engine.knrm_forward(q, d, packed, V, D, mu, sigma, a1, b1, w2, b2, scoretanh=bool(hidden)).cpu().numpy()
        lists = engine.knrm_forward_lists(off, packed, V, D, mu, sigma, a1, b1, w2, b2, scoretanh=bool(hidden), query=q, doc=d).cpu().numpy()
        scale = float(np.abs(pairwise).max())
        assert np.abs(lists - pairwise).max() <= 2e-5 * scale, (K, hidden, np.abs(lists - pairwise).max(), scale)


@pytest.mark.parametrize("cfg,Q", [({}, 4), ({"crossmatch": False, "singlefc": False, "scoretanh": True}, 4), ({"maxngram": 2, "filters": 64}, 3),
                                   ({}, 7), ({"maxngram": 1, "filters": 32}, 2)])
def test_convknrm_lists_are_bit_identical_to_the_per_pair_kernel(cfg, Q):
    """ConvKNRM over whole lists (capamd_convknrm_forward_lists): the unigram document view computed once per distinct token of a list -
    the same normalisation, f16 split and matrix instructions as the per-pair kernel's phase B - and looked up per position; the other
    n-gram views and the pooling unchanged: scores equal capamd_convknrm_forward's bit for bit.  Ragged lists (1 .. 300 documents), pads,
    an all-pad document, more than 8 lists; ids as rows and through a candidate store; cross-match and not, 1-3 n-gram sizes, a query of
    seven terms (two blocks of sixteen query vectors)."""
    from capreolus_amd.feeder import CandidateStore
    from capreolus_amd.reranker import ConvKNRM

    V, D = 3000, 300
    emb = synthetic.make_embeddings(V, D, seed=11)
    docs = [300, 1, 9, 70, 2, 5, 33, 4, 12, 8]
    b, off = _lists_batch(len(docs), docs, V, 43, Q=Q, L=200)
    b["query"] = np.clip(b["query"], 0, None)          # (ConvKNRM takes no OOV ids: nn.Embedding would raise)
    b["posdoc"] = np.clip(b["posdoc"], 0, None)
    b["posdoc"][7] = 0
    r = ConvKNRM(cfg, SimpleNamespace(embeddings=emb, config={"maxqlen": Q}, pad=0))
    torch.manual_seed(5)
    r.build_model().to(DEV).eval()
    d = {k: _t(v) for k, v in b.items()}
    with torch.no_grad():
        pairwise = r.test(d)
        lists = r.test_lists(d, off)
        assert torch.isfinite(pairwise).all() and float(pairwise.abs().max()) > 0
        assert torch.equal(pairwise, lists), float((pairwise - lists).abs().max())
        # the documents of every list in another order; lists cut in two
        perm = np.concatenate([off[k] + np.random.RandomState(k).permutation(docs[k]) for k in range(len(docs))])
        assert torch.equal(r.test_lists({k: v[perm] for k, v in d.items()}, off), pairwise[perm])
        cuts = np.unique(np.concatenate([off, off[:-1] + np.maximum(1, np.asarray(docs) // 2)]))
        assert torch.equal(r.test_lists(d, cuts), pairwise)
        store = CandidateStore(DEV)
        pq, pd = [], []
        for l, n in enumerate(docs):
            for i in range(off[l], off[l] + n):
                pq.append(store.add_query(f"q{l}", b["query"][off[l]], b["query_idf"][off[l]]))
                pd.append(store.add_doc(f"d{i}", b["posdoc"][i]))
        store.finalize()
        assert torch.equal(r.test_resident_lists(store, _t(np.asarray(pq, np.int32)), _t(np.asarray(pd, np.int32)), off), pairwise)
        # a list whose pairs bring different query rows is refused, as for the other list entries
        bad = {k: v.clone() for k, v in d.items()}
        bad["query"][off[3] + 1, 0] = (bad["query"][off[3] + 1, 0] % (V - 2)) + 1 if bad["query"][off[3] + 1, 0] != 1 else 2
        with pytest.raises(ValueError):
            r.test_lists(bad, off)


@pytest.mark.parametrize("kind,Q", [("knrm", 4), ("drmm", 4), ("drmmtks", 4), ("pacrr", 4), ("knrm", 8), ("drmm", 6), ("drmmtks", 5), ("knrm", 5)])
def test_full_size_list_properties(kind, Q):
    """The list route at BASELINE.json's geometry (vocabulary 400,001 x 300, 800-term documents, 16 queries x 1000 candidates): a
    document's score does not depend on where it stands in its list or on how the query's documents are cut into lists (both bit for
    bit: the table of a (query, term) pair holds the same four similarities whichever list computes it, and a document is pooled alone),
    and the scores equal the per-pair kernels' - bit for bit for DRMM / DRMM-TKS / PACRR, to fp32 rounding of the sums for KNRM."""
    from capreolus_amd.reranker import DRMMTKS, PACRR

    V, D, NQ, ND = 400001, 300, 16, 1000
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    emb = torch.randn((V, D), generator=g, device=DEV) * 0.4
    emb[0] = 0
    batch = synthetic.make_candidate_list_torch(NQ, ND, V, DEV, seed=5, maxqlen=Q)
    if kind != "knrm":
        batch = {**batch, "query": batch["query"].clamp(min=0)}
    torch.manual_seed(1)
    ext = SimpleNamespace(embeddings=np.zeros((2, 300), dtype=np.float32), config={"maxqlen": Q}, pad=0)
    r = {"knrm": KNRM, "drmm": DRMM, "drmmtks": DRMMTKS, "pacrr": PACRR}[kind]({}, ext)
    m = r.build_model().to(DEV).eval()
    m.embedding = torch.nn.Embedding.from_pretrained(emb, freeze=True)
    off = np.arange(0, NQ * ND + 1, ND, dtype=np.int64)
    with torch.no_grad():
        s = r.test_lists(batch, off)
        assert s.shape == (NQ * ND,) and torch.isfinite(s).all()
        pairwise = r.test(batch)
        if kind == "knrm":
            assert (s - pairwise).abs().max() <= 2e-5 * pairwise.abs().max()
        else:
            assert torch.equal(s, pairwise)
        # (a) the documents of every list in another order
        perm = torch.cat([q * ND + torch.randperm(ND, device=DEV) for q in range(NQ)])
        assert torch.equal(r.test_lists({k: v[perm] for k, v in batch.items()}, off), s[perm])
        # (b) every query's documents cut into three lists of unequal length (48 lists: one launch group), and the whole run in groups
        #     of 5 lists per call (the last call has one list)
        cuts = np.sort(np.concatenate([off, off[:-1] + 1, off[:-1] + 377]))
        assert torch.equal(r.test_lists(batch, cuts), s)
        parts = [r.test_lists({k: v[lo * ND:hi * ND] for k, v in batch.items()}, off[: hi - lo + 1]) for lo, hi in
                 [(i, min(i + 5, NQ)) for i in range(0, NQ, 5)]]
        assert torch.equal(torch.cat(parts), s)


@pytest.mark.parametrize("topk", [10, 3, 16])
def test_drmmtks_lists_are_bit_identical_to_the_per_pair_kernel(topk):
    """DRMM-TKS over whole lists (capamd_drmmtks_forward_lists): top-k selections of bit-identical similarities, fed to the Linear in the
    same order - the scores equal capamd_drmmtks_forward's bit for bit; ragged lists, OOV query terms with exact matches in the
    documents, an all-pad document, a document of fewer real terms than k; ids as rows and through a candidate store."""
    from capreolus_amd.feeder import CandidateStore
    from capreolus_amd.reranker import DRMMTKS

    V, D = 3000, 300
    emb = synthetic.make_embeddings(V, D, seed=8)
    docs = [300, 1, 17, 90]
    b, off = _lists_batch(len(docs), docs, V, 31)
    b["posdoc"][7] = 0
    b["posdoc"][8, 5:] = 0
    b["posdoc"][9, 3] = b["query"][9, 0] if b["query"][9, 0] < 0 else -4
    r = DRMMTKS({"topk": topk}, SimpleNamespace(embeddings=emb))
    torch.manual_seed(2)
    m = r.build_model().to(DEV).eval()
    with torch.no_grad():
        m.gates.weight.mul_(40.0)
        m.ffw[0].weight.mul_(6.0)
    d = {k: _t(v) for k, v in b.items()}
    with torch.no_grad():
        pairwise = r.test(d)
        lists = r.test_lists(d, off)
    assert torch.equal(lists, pairwise), float((lists - pairwise).abs().max())
    store = CandidateStore(DEV)
    for i in range(len(off) - 1):
        store.add_query(f"q{i}", b["query"][off[i]], b["query_idf"][off[i]])
    for j in range(b["posdoc"].shape[0]):
        store.add_doc(f"d{j}", b["posdoc"][j])
    store.finalize()
    pq = _t(np.repeat(np.arange(len(docs)), docs).astype(np.int32))
    pd = _t(np.arange(b["posdoc"].shape[0], dtype=np.int32))
    with torch.no_grad():
        assert torch.equal(r.test_resident_lists(store, pq, pd, off), pairwise)


@pytest.mark.parametrize("cfg", [{}, {"kmax": 4, "nfilters": 16, "idf": False, "nonlinearity": "tanh"}, {"mingram": 2, "maxgram": 2}])
def test_pacrr_lists_are_bit_identical_to_the_per_pair_kernel(cfg):
    """PACRR over whole lists (capamd_pacrr_forward_lists): the MFMA kernel with a table lookup per position as its front end - the
    similarity matrix, and with it the score, equals capamd_pacrr_forward's bit for bit; ragged lists, OOV query terms with exact matches
    in the documents, an all-pad document; ids as rows and through a candidate store."""
    from capreolus_amd.feeder import CandidateStore
    from capreolus_amd.reranker import PACRR

    V, D = 3000, 300
    emb = synthetic.make_embeddings(V, D, seed=9)
    docs = [200, 1, 9, 70]
    b, off = _lists_batch(len(docs), docs, V, 41)
    b["posdoc"][7] = 0
    b["posdoc"][9, 3] = b["query"][9, 0] if b["query"][9, 0] < 0 else -4
    r = PACRR(cfg, SimpleNamespace(embeddings=emb, config={"maxqlen": 4}, pad=0))
    torch.manual_seed(4)
    r.build_model().to(DEV).eval()
    assert r.supports_lists
    d = {k: _t(v) for k, v in b.items()}
    with torch.no_grad():
        pairwise = r.test(d)
        lists = r.test_lists(d, off)
    assert torch.equal(lists, pairwise), float((lists - pairwise).abs().max())
    store = CandidateStore(DEV)
    for i in range(len(off) - 1):
        store.add_query(f"q{i}", b["query"][off[i]], b["query_idf"][off[i]])
    for j in range(b["posdoc"].shape[0]):
        store.add_doc(f"d{j}", b["posdoc"][j])
    store.finalize()
    pq = _t(np.repeat(np.arange(len(docs)), docs).astype(np.int32))
    pd = _t(np.arange(b["posdoc"].shape[0], dtype=np.int32))
    with torch.no_grad():
        assert torch.equal(r.test_resident_lists(store, pq, pd, off), pairwise)


@pytest.mark.parametrize("D,Q,L,V,docs", [
    (50, 3, 37, 700, [5, 1, 40]),                       # one float4 chunk per lane, three query terms, a short odd document length
    (100, 1, 130, 1500, [9] * 11),                      # one query term; more than 8 lists (the XCD-aware numbering)
    (200, 4, 1000, 2100, [2, 300]),                     # documents longer than the benchmark's; ids beyond one 1024-id block
    (300, 2, 64, 1030, [3] * 70),                       # more lists than one launch group holds (64)
    (300, 5, 200, 2500, [30, 7, 120]),                  # five query terms: a second block of ONE term (`maxqlen` is a free option)
    (100, 6, 90, 1200, [11] * 9),                       # six terms, more than 8 lists
    (300, 8, 800, 3000, [250, 40]),                     # eight terms at the benchmark's document length
    (50, 7, 33, 600, [1, 2, 3, 50]),                    # seven terms, single-document lists
])
def test_lists_random_geometries(D, Q, L, V, docs):
    """The list route over the shapes the per-pair sweep covers: embedding widths of 1-5 float4 chunks per lane, 1-8 query terms (two
    blocks of four from the fifth on), short / long documents, fewer and more lists than an XCD group and than a launch group - KNRM
    against the per-pair kernel to fp32 rounding of its sums, DRMM and DRMM-TKS bit for bit."""
    emb = synthetic.make_embeddings(V, D, seed=D + Q)
    b, off = _lists_batch(len(docs), docs, V, 100 + L, Q=Q, L=L)
    d = {k: _t(v) for k, v in b.items()}
    r = KNRM({}, SimpleNamespace(embeddings=emb))
    torch.manual_seed(L)
    r.build_model().to(DEV).eval()
    with torch.no_grad():
        pairwise, lists = r.test(d).cpu().numpy(), r.test_lists(d, off).cpu().numpy()
    scale = float(np.abs(pairwise).max())
    assert np.abs(lists - pairwise).max() <= 2e-5 * scale, (np.abs(lists - pairwise).max(), scale)
    no_oov = {**d, "query": d["query"].clamp(min=0)}      # (DRMM refuses OOV query terms)
    r = DRMM({}, SimpleNamespace(embeddings=emb))
    torch.manual_seed(L)
    r.build_model().to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(r.test_lists(no_oov, off), r.test(no_oov))
        for hist, gate in (("NH", "IDF"), ("LCH", "TV")):
            r2 = DRMM({"histType": hist, "gateType": gate}, SimpleNamespace(embeddings=emb))
            torch.manual_seed(L + 1)
            r2.build_model().to(DEV).eval()
            assert torch.equal(r2.test_lists(no_oov, off), r2.test(no_oov)), (hist, gate)
    from capreolus_amd.reranker import DRMMTKS

    if L >= 10:
        r = DRMMTKS({"topk": 10}, SimpleNamespace(embeddings=emb, config={"maxqlen": Q}, pad=0))
        torch.manual_seed(L)
        r.build_model().to(DEV).eval()
        with torch.no_grad():
            assert torch.equal(r.test_lists(no_oov, off), r.test(no_oov))


@pytest.mark.parametrize("case", ["same", "disjoint", "odd", "alone", "half"])
def test_lists_sims_neighbouring_lists_row_classes(case):
    """The sims pass on neighbouring lists built class by class (written for round 6's two-lists-per-workgroup kernel, which was measured
    and removed - profiles/r06/lists_sims_pairs_ab.txt; kept for the one-list kernel): the same vocabulary in both lists of a pair,
    disjoint vocabularies, odd counts of shared / unshared rows per 1024-id block (an odd last row is done twice), an odd number of lists,
    and neighbouring lists of which only ONE has a two-term query (the two-term form of the dot products).  DRMM and DRMM-TKS must equal
    their per-pair kernels bit for bit (the table entries are the per-pair kernels' similarities), KNRM to fp32 rounding of its sums."""
    from capreolus_amd.reranker import DRMMTKS

    V, D, Q, L = 2300, 300, 4, 60
    emb = synthetic.make_embeddings(V, D, seed=11)
    rs = np.random.RandomState({"same": 1, "disjoint": 2, "odd": 3, "alone": 4, "half": 5}[case])
    n_lists, n_docs = (3 if case == "alone" else 4), 7
    if case == "same":
        vocab = [np.arange(1, V)] * n_lists
    elif case == "disjoint":
        vocab = [np.arange(1 + k, V, n_lists) for k in range(n_lists)]
    elif case == "odd":      # per 1024-id block: 3 ids in both lists of a pair, 5 only in the first, 1 only in the second
        both, first, second = [10, 500, 1023, 1030, 1500, 2047, 2050, 2100, 2299], [1, 2, 3, 4, 5, 1100, 1101, 1102, 1103, 1104, 2200, 2201, 2202, 2203, 2204], [700, 1700, 2250]
        vocab = [np.array(both + first), np.array(both + second)] * 2
    else:
        vocab = [rs.choice(np.arange(1, V), size=400, replace=False) for _ in range(n_lists)]
    q = np.zeros((n_lists * n_docs, Q), np.int64)
    d = np.zeros((n_lists * n_docs, L), np.int64)
    idf = np.zeros((n_lists * n_docs, Q), np.float32)
    for k in range(n_lists):
        nq = 2 if (case == "half" and k % 2 == 0) else 1 + (k + 2) % Q
        qk = np.zeros(Q, np.int64)
        qk[:nq] = rs.choice(vocab[k], size=nq)
        ik = np.where(qk != 0, rs.uniform(0.5, 8.0, size=Q), 0.0).astype(np.float32)
        for j in range(n_docs):
            n = rs.randint(1, L + 1)
            q[k * n_docs + j], idf[k * n_docs + j] = qk, ik
            d[k * n_docs + j, :n] = rs.choice(vocab[k], size=n)
    if case == "odd":        # every id of the class sets must occur
        for k in range(n_lists):
            d[k * n_docs, : len(vocab[k])] = vocab[k]
    off = np.arange(0, n_lists * n_docs + 1, n_docs).astype(np.int64)
    t = {"query": _t(q), "posdoc": _t(d), "query_idf": _t(idf)}
    r = DRMM({}, SimpleNamespace(embeddings=emb))
    torch.manual_seed(2)
    r.build_model().to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(r.test_lists(t, off), r.test(t))
    r = DRMMTKS({"topk": 5}, SimpleNamespace(embeddings=emb, config={"maxqlen": Q}, pad=0))
    torch.manual_seed(2)
    r.build_model().to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(r.test_lists(t, off), r.test(t))
    r = KNRM({}, SimpleNamespace(embeddings=emb))
    torch.manual_seed(2)
    r.build_model().to(DEV).eval()
    with torch.no_grad():
        pairwise, lists = r.test(t).cpu().numpy(), r.test_lists(t, off).cpu().numpy()
    assert np.abs(lists - pairwise).max() <= 2e-5 * float(np.abs(pairwise).max())


@pytest.mark.parametrize("Q", [5, 6, 8])
def test_lists_with_long_queries_through_predict(Q):
    """`PytorchTrainer.predict` on a run whose extractor was configured with `maxqlen` > 4 (reference extractor/embedtext.py:28-31): the
    resident route still scores whole lists (two blocks of four query terms), DRMM's predictions equal the DataLoader route's bit for bit,
    KNRM's to fp16 rounding of sums that differ by 1e-6; short queries under the long `maxqlen` (pads in block 0, an empty block 1) too."""
    from capreolus_amd.trainer import PytorchTrainer

    V, L = 2000, 120
    emb = synthetic.make_embeddings(V, 300, seed=Q)
    docs = [40, 25, 60]
    b, off = _lists_batch(len(docs), docs, V, 50 + Q, Q=Q, L=L)
    b["query"] = np.clip(b["query"], 0, None)
    b["query"][off[1]:off[2], 2:] = 0          # the second list's query: two terms under maxqlen = Q
    q2d = {str(k): [f"d{i}" for i in range(off[k], off[k + 1])] for k in range(len(docs))}

    class Sampler(torch.utils.data.IterableDataset):
        qid_to_docids = q2d

        def __iter__(self):
            for qid, ds in q2d.items():
                for d in ds:
                    i = int(d[1:])
                    yield {"qid": qid, "posdocid": d, "query": b["query"][i], "posdoc": b["posdoc"][i], "query_idf": b["query_idf"][off[int(qid)]]}

        def __len__(self):
            return int(off[-1])

        def get_qid_docid_pairs(self):
            return ((q, d) for q, ds in q2d.items() for d in ds)

    for cls in (KNRM, DRMM):
        r = cls({}, SimpleNamespace(embeddings=emb))
        torch.manual_seed(Q)
        r.build_model().to(DEV).eval()
        calls = []
        real = cls.test_resident_lists
        cls.test_resident_lists = lambda self, *a, _real=real: calls.append(1) or _real(self, *a)
        try:
            got = PytorchTrainer({"batch": 32}).predict(r, Sampler())
        finally:
            cls.test_resident_lists = real
        assert calls, "the list route should have taken this run"
        want = PytorchTrainer({"batch": 32, "resident": False}).predict(r, Sampler())
        if cls is DRMM:
            assert got == want
        else:
            flat = lambda p: np.array([p[q][d] for q, ds in q2d.items() for d in ds], dtype=np.float64)      # noqa: E731
            assert np.abs(flat(got) - flat(want)).max() <= 2e-3 * np.abs(flat(want)).max()


@pytest.mark.parametrize("model", ["knrm", "drmm"])
def test_predict_scores_whole_lists_where_the_reranker_can(model, monkeypatch):
    """`PytorchTrainer.predict` on its resident route hands whole candidate lists to every reranker that takes them (`lists` = "always", the
    default) or only to those whose list scores equal their per-pair scores bit for bit (`lists` = "exact": DRMM, not KNRM): same
    predictions as the DataLoader route on the reference's 200-candidate ranking list."""
    from capreolus_amd.trainer import PytorchTrainer

    c = load_case(model, "ranklist")
    r = _knrm_model(c) if model == "knrm" else _drmm_model(c)
    B = c["query"].shape[0]
    q2d = {"5": [f"d{i}" for i in range(0, 120)], "6": [f"d{i}" for i in range(120, B)]}

    class Sampler(torch.utils.data.IterableDataset):
        qid_to_docids = q2d

        def __iter__(self):
            for qid, docs in q2d.items():
                for dd in docs:
                    yield {"qid": qid, "posdocid": dd, "query": c["query"][0], "posdoc": c["posdoc"][int(dd[1:])], "query_idf": c["query_idf"][0]}

        def __len__(self):
            return B

        def get_qid_docid_pairs(self):
            return ((q, dd) for q, docs in q2d.items() for dd in docs)

    calls = []
    real = type(r).test_resident_lists
    monkeypatch.setattr(type(r), "test_resident_lists", lambda self, *a: calls.append(1) or real(self, *a))
    s = Sampler()
    want = PytorchTrainer({"batch": 32, "resident": False}).predict(r, s)
    got = PytorchTrainer({"batch": 32, "lists": "exact"}).predict(r, s)
    assert got == want and len(calls) == (1 if model == "drmm" else 0)       # "exact": DRMM as lists, KNRM through the per-pair kernel
    got = PytorchTrainer({"batch": 32}).predict(r, s)                        # the default: "always"
    assert len(calls) == (2 if model == "drmm" else 1)
    assert got == want       # (KNRM: equal on this list - its scores sit >= 6.9e-6 from an fp16 rounding boundary - not by construction)
    assert PytorchTrainer({"batch": 32, "lists": "never"}).predict(r, s) == want and len(calls) == (2 if model == "drmm" else 1)
    if model == "knrm":
        flat = np.array([got[q][dd] for q, docs in q2d.items() for dd in docs], dtype=np.float16)
        assert np.array_equal(flat, c["ref_scores_f16"])
