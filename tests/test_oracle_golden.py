"""Pins the CPU oracle (oracle/interaction_oracle.c) against golden vectors produced by the
REFERENCE nn.Modules (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import cpu as oracle
from tests.helpers import DRMM_CASES, KNRM_CASES, REL_TOL, knrm_weights, load_case, rank_order, rel_err


@pytest.mark.parametrize("name", KNRM_CASES)
def test_knrm_oracle_matches_reference(name):
    c = load_case("knrm", name)
    packed = oracle.pack(c["emb"])
    mu, sigma, w1, b1, w2, b2 = knrm_weights(c)
    got, err = oracle.knrm(c["query"], c["posdoc"], packed, int(c["D"]), mu, sigma, w1, b1, w2, b2, bool(c["scoretanh"]))
    assert err == 0
    e = rel_err(got, c["ref_scores"])
    assert e.max() <= REL_TOL, (name, e.max(), int(e.argmax()))
    # the interaction matrix itself: row sums of the reference's simmat
    sim, err = oracle.simmat(c["query"], c["posdoc"], packed, int(c["D"]))
    assert err == 0
    np.testing.assert_allclose(sim.sum(axis=2, dtype=np.float64), c["ref_sim_rowsum"], rtol=1e-4, atol=2e-4)


def test_knrm_oracle_rank_order_matches_reference():
    c = load_case("knrm", "ranklist")
    packed = oracle.pack(c["emb"])
    mu, sigma, w1, b1, w2, b2 = knrm_weights(c)
    got, _ = oracle.knrm(c["query"], c["posdoc"], packed, int(c["D"]), mu, sigma, w1, b1, w2, b2, bool(c["scoretanh"]))
    # all 200 fp16 scores are the reference's bit for bit (its scores sit >= 6.9e-6 relative from an fp16 rounding boundary, the
    # oracle within 5e-7 of them): the same run order, outright
    assert np.array_equal(got.astype(np.float16), c["ref_scores_f16"])
    assert np.array_equal(rank_order(got.astype(np.float16)), rank_order(c["ref_scores_f16"]))


def _drmm_run(c):
    packed = oracle.pack(c["emb"])
    return oracle.drmm(
        c["query"], c["posdoc"], c["query_idf"], packed, int(c["D"]), c["edges"], str(c["histType"]), str(c["gateType"]),
        c["sd.gates.weight"], c["emb"], c["sd.ffw.0.weight"], c["sd.ffw.0.bias"], c["sd.ffw.2.weight"], c["sd.ffw.2.bias"],
        c["sd.output_layer.weight"], c["sd.output_layer.bias"])


@pytest.mark.parametrize("name", DRMM_CASES)
def test_drmm_oracle_matches_reference(name):
    c = load_case("drmm", name)
    got, counts, err = _drmm_run(c)
    assert err == 0
    amb = c["n_ambiguous"]
    nb = int(c["nbins"])
    d = counts.astype(np.int64) - c["ref_counts"].astype(np.int64)
    diff = np.abs(d).sum(axis=(1, 2))
    # (1) histogram counts: identical wherever no similarity sits within 4 ulp of a bin edge.  Where
    # some do -- SURVEY.md §7: identical in-vocab terms give cos in {1-ulp, 1, 1+ulp} depending on
    # the BLAS summation order, so the reference itself flips a coin on "cos < 1.0" -- at most that
    # many counts may move, and only in the last regular bin [edge_{nbins-2}, 1.0).
    assert (diff[amb == 0] == 0).all(), (name, np.nonzero((diff > 0) & (amb == 0))[0])
    assert (diff <= amb).all(), (name, diff, amb)
    moved = np.nonzero(np.abs(d).sum(axis=(0, 1)))[0]
    assert set(moved.tolist()) <= {nb - 1}, (name, moved)
    # (2) scores: within 1e-3 wherever the counts agree (that covers every unambiguous pair)
    e = rel_err(got, c["ref_scores"])
    clean = diff == 0
    assert clean.sum() >= 0.3 * len(clean), name
    assert e[clean].max() <= REL_TOL, (name, e[clean].max())
    # (3) where counts moved the score moves with them and nothing else: bounded by what the moved
    # counts can do through log(count+1) and the tanh MLP (loose bound, reported in DESIGN.md)
    assert e.max() <= 0.05, (name, e.max())


@pytest.mark.parametrize("name", DRMM_CASES)
def test_drmm_back_end_on_reference_counts(name):
    """the oracle's histogram -> log/normalise -> ffw -> gate -> output back end, fed with the REFERENCE's
    own raw bin counts, reproduces the reference's scores on EVERY pair - so the back end is pinned independently of the
    cos(a, a) < 1.0 coin flip that moves counts in the front end."""
    c = load_case("drmm", name)
    got, err = oracle.drmm_from_counts(c["ref_counts"], c["query"], c["query_idf"], int(c["V"]), int(c["D"]), str(c["histType"]),
                                       str(c["gateType"]), c["sd.gates.weight"], c["emb"], c["sd.ffw.0.weight"], c["sd.ffw.0.bias"],
                                       c["sd.ffw.2.weight"], c["sd.ffw.2.bias"], c["sd.output_layer.weight"], c["sd.output_layer.bias"])
    assert err == 0
    e = rel_err(got, c["ref_scores"])
    assert e.max() <= 1e-5, (name, e.max(), int(e.argmax()))
    # and fed with the oracle's own counts it is the oracle (same function, sanity of the plumbing)
    full, counts, _ = _drmm_run(c)
    again, _ = oracle.drmm_from_counts(counts, c["query"], c["query_idf"], int(c["V"]), int(c["D"]), str(c["histType"]), str(c["gateType"]),
                                       c["sd.gates.weight"], c["emb"], c["sd.ffw.0.weight"], c["sd.ffw.0.bias"], c["sd.ffw.2.weight"],
                                       c["sd.ffw.2.bias"], c["sd.output_layer.weight"], c["sd.output_layer.bias"])
    assert np.array_equal(full, again)


# What the cos(a, a) < 1.0 coin flip (DRMM.py:62-66 on an ulp-level similarity) costs against the reference, per fixture, measured
# and bounded here and quoted in DESIGN.md section 1:  (#pairs with a term within 4 ulp of an edge, #pairs whose counts moved,
# #counts moved, max relative score delta, fraction of pairs beyond 1e-3).  The bounds are the measured values rounded up.
DRMM_COIN_FLIP_BOUNDS = {
    "default": dict(pairs_moved=13, counts_moved=376, max_delta=1.2e-2, frac_over=0.46),     # (one pair repeats a query term 136 times)
    "zero_idf": dict(pairs_moved=0, counts_moved=0, max_delta=1e-6, frac_over=0.0),
    "tv_nh": dict(pairs_moved=4, counts_moved=7, max_delta=9.1e-3, frac_over=0.25),
    "ch": dict(pairs_moved=5, counts_moved=57, max_delta=1e-5, frac_over=0.0),
    "ranklist": dict(pairs_moved=78, counts_moved=168, max_delta=1.04e-2, frac_over=0.365),   # 200 candidates of one query
}


@pytest.mark.parametrize("name", DRMM_CASES)
def test_drmm_coin_flip_statistics(name):
    c = load_case("drmm", name)
    got, counts, _ = _drmm_run(c)
    d = np.abs(counts.astype(np.int64) - c["ref_counts"].astype(np.int64))
    e = rel_err(got, c["ref_scores"])
    stats = dict(pairs=len(got), pairs_ambiguous=int((c["n_ambiguous"] > 0).sum()), pairs_moved=int((d.sum(axis=(1, 2)) > 0).sum()),
                 counts_moved=int(d.sum()), max_delta=float(e.max()), frac_over=float((e > REL_TOL).mean()))
    print("DRMM coin flip", name, stats)
    b = DRMM_COIN_FLIP_BOUNDS[name]
    assert stats["pairs_moved"] <= b["pairs_moved"] and stats["counts_moved"] <= b["counts_moved"], stats
    assert stats["max_delta"] <= b["max_delta"] and stats["frac_over"] <= b["frac_over"], stats


# Where the reference reproduces ITSELF bit for bit under every blocking tried (below), this build's moved counts are not the reference's
# noise but a genuine disagreement of two summation orders on cos(a, a) vs 1.0: frozen here, reported in DESIGN.md section 1.
DRMM_GENUINE_DISAGREEMENT = {
    "tv_nh": dict(pairs_moved=4, counts_moved=7, max_delta=9.02e-3, frac_over=0.25),     # 4 of 16 pairs beyond 1e-3 (NH normalises the counts)
    "ch": dict(pairs_moved=5, counts_moved=57, max_delta=3.4e-6, frac_over=0.0),         # counts move, scores do not (CH feeds raw counts to a saturated tanh)
}


def _fp16_rank_inversions(got, want):
    """pairs of candidates (i, j) that the two fp16-rounded score lists order differently (ties keep first-stage order on both sides)"""
    a, b = rank_order(np.asarray(got).astype(np.float16)), rank_order(np.asarray(want).astype(np.float16))
    pos_b = np.empty(len(b), dtype=np.int64)
    pos_b[b] = np.arange(len(b))
    seq = pos_b[a]
    return int(sum(int((seq[i + 1:] < seq[i]).sum()) for i in range(len(seq))))


@pytest.mark.parametrize("name", ["default", "ranklist", "tv_nh", "ch"])
def test_drmm_coin_flip_is_the_references_own_noise(name):
    """The reference against ITSELF (tests/golden/make_golden_extra.py: the same module, weights and inputs under ten other `bmm` blockings -
    one pair per call, chunks of 2 / 4 / 8 / 32 pairs, 1 / 2 / 4 threads, the batch reversed, denormals flushed) on the `sim < 1.0` counts
    of DRMM.py:62-66.  `default` / `ranklist`: its batch-of-one run moves MORE counts against its own batched run than this build's
    documented summation order does against the fixture - the oracle's (= the GPU kernel's: counts are asserted bit-exact between them)
    distance to the reference is bounded by the reference's distance to its own second run, measure by measure.  `tv_nh` / `ch` (D = 50 /
    100): the reference reproduces itself under all ten, so what this build moves there (4 and 5 pairs) is a GENUINE disagreement of
    summation orders on cos(a, a) against 1.0 - asserted as the frozen numbers of DRMM_GENUINE_DISAGREEMENT, not skipped."""
    import os

    from tests.helpers import GOLDEN

    c = load_case("drmm", name)
    alt = np.load(os.path.join(GOLDEN, f"drmm_{name}_alt.npz"))
    assert bool(alt["regenerates_fixture"])            # the generator reproduced drmm_<name>.npz before it varied the blocking
    got, counts, _ = _drmm_run(c)

    def distance(scores, cnt):
        d = np.abs(cnt.astype(np.int64) - c["ref_counts"].astype(np.int64))
        e = rel_err(scores, c["ref_scores"])
        return dict(pairs_moved=int((d.sum(axis=(1, 2)) > 0).sum()), counts_moved=int(d.sum()), max_delta=float(e.max()),
                    frac_over=float((e > REL_TOL).mean()))

    ours = distance(got, counts)
    variants = [str(v) for v in alt["variants"]]
    assert len(variants) >= 10
    ref_self = {k: distance(alt[k + "_scores"], alt[k + "_counts"]) for k in variants}
    worst = {m: max(v[m] for v in ref_self.values()) for m in ours}
    print("DRMM reference vs itself", name, {k: v for k, v in ref_self.items() if v["pairs_moved"]}, "worst", worst, "this build", ours,
          "fp16 rank inversions vs the reference (whole batch as one list):", _fp16_rank_inversions(got, c["ref_scores"]))
    if worst["pairs_moved"] == 0:
        g = DRMM_GENUINE_DISAGREEMENT[name]
        assert ours["pairs_moved"] == g["pairs_moved"] and ours["counts_moved"] == g["counts_moved"], (ours, g)
        assert ours["max_delta"] <= g["max_delta"] and abs(ours["frac_over"] - g["frac_over"]) < 1e-9, (ours, g)
        return
    assert ours["pairs_moved"] <= worst["pairs_moved"] and ours["counts_moved"] <= worst["counts_moved"], (ours, worst)
    assert ours["max_delta"] <= worst["max_delta"] * 1.05 and ours["frac_over"] <= worst["frac_over"], (ours, worst)


def test_drmm_ranklist_rank_inversions_against_the_reference():
    """The 200-candidate DRMM ranking list: how far the run order moves against the reference's (fp16-rounded scores, stable order) - and
    against the reference's own batch-of-one run, which moves it further.  Frozen counts of inverted candidate pairs."""
    import os

    from tests.helpers import GOLDEN

    c = load_case("drmm", "ranklist")
    alt = np.load(os.path.join(GOLDEN, "drmm_ranklist_alt.npz"))
    got, _, _ = _drmm_run(c)
    ours = _fp16_rank_inversions(got, c["ref_scores"])
    ref_batch1 = _fp16_rank_inversions(alt["batch1_scores"], c["ref_scores"])
    print("DRMM ranklist: inverted pairs of 19,900 - this build", ours, "the reference's batch-of-one run", ref_batch1)
    assert ours <= ref_batch1
    assert ours <= 1500, ours          # (measured 1,495; the reference against its own batch-of-one run: 2,236)


def test_drmm_oracle_rejects_oov_query():
    c = load_case("drmm", "default")
    c["query"] = c["query"].copy()
    c["query"][0, 0] = -3
    _, _, err = _drmm_run(c)
    assert err & 4  # reference: IndexError at DRMM.py:109


@pytest.mark.parametrize("name", KNRM_CASES)
def test_knrm_torch_port_matches_reference(name):
    import torch

    from oracle import torch_port

    c = load_case("knrm", name)
    mu, sigma, w1, b1, w2, b2 = (None if x is None else torch.as_tensor(x) for x in knrm_weights(c))
    with torch.no_grad():
        got = torch_port.knrm(torch.as_tensor(c["emb"]), torch.as_tensor(c["query"]), torch.as_tensor(c["posdoc"]), mu, sigma,
                              w1, b1, w2, b2, bool(c["scoretanh"])).numpy()
    assert rel_err(got, c["ref_scores"]).max() <= 1e-5


@pytest.mark.parametrize("name", DRMM_CASES)
def test_drmm_torch_port_matches_reference(name):
    import torch

    from oracle import torch_port

    c = load_case("drmm", name)
    t = lambda k: torch.as_tensor(c[k])  # noqa: E731
    with torch.no_grad():
        got = torch_port.drmm(t("emb"), t("query"), t("posdoc"), t("query_idf"), int(c["nbins"]), str(c["histType"]),
                              str(c["gateType"]), t("sd.gates.weight"), t("sd.ffw.0.weight"), t("sd.ffw.0.bias"),
                              t("sd.ffw.2.weight"), t("sd.ffw.2.bias"), t("sd.output_layer.weight"),
                              t("sd.output_layer.bias")).numpy()
    # same ATen ops in the same order as the reference -> same coin flips on cos(a,a) < 1
    assert rel_err(got, c["ref_scores"]).max() <= 1e-5


@pytest.mark.parametrize("name", ["mini", "mini_s128", "base", "base_long"])
def test_bert_port_matches_reference(name):
    from oracle import bert_port
    from tests.helpers import load_bert_case

    c = load_bert_case(name)
    if name.startswith("base"):  # 12 layers x 12 passages on CPU: keep the CPU suite short -> score the first document only
        sl = slice(0, 1)
    else:
        sl = slice(None)
    inp, mask, seg = (c[k][sl] for k in ("pos_bert_input", "pos_mask", "pos_seg"))
    for agg in ("max", "first", "sum") if name.startswith("base") else ("max", "first", "sum", "avg"):   # (avg: batch-wide denominator)
        got = bert_port.maxp(c["weights"], inp, mask, seg, c["heads"], c["layers"], agg).numpy()
        assert rel_err(got, c["ref_" + agg][sl]).max() <= 2e-5, (name, agg)


@pytest.mark.parametrize("name", ["roberta_mini", "roberta_h256"])
def test_roberta_port_matches_reference(name):
    """RoBERTa bodies behind ptBERTMaxP (ptBERTMaxP.py:46-48, 57-58): position ids counted over non-pad tokens, LayerNorm eps 1e-5, the
    `dense -> tanh -> out_proj` head, token types zeroed - whence `sum` scores 0 and `avg` is 0/0 in the reference."""
    import torch

    from oracle import bert_port
    from tests.helpers import load_roberta_case

    c = load_roberta_case(name)
    w = bert_port.roberta_as_bert(c["weights"])
    zeros = torch.zeros_like(c["pos_seg"])
    for agg in ("max", "first", "sum", "avg"):
        got = bert_port.maxp(w, c["pos_bert_input"], c["pos_mask"], zeros, c["heads"], c["layers"], agg, eps=1e-5, pos_pad_id=1).numpy()
        ref = c["ref_" + agg]
        if agg == "avg":
            assert np.isnan(ref).all() and np.isnan(got).all()
        elif agg == "sum":
            assert (ref == 0).all() and (got == 0).all()
        else:
            assert rel_err(got, ref).max() <= 2e-5, (name, agg)


@pytest.mark.parametrize("kind", ["knrm", "drmm"])
def test_ndcg20_parity_oracle_vs_reference(kind):
    """The metric's parity half (BASELINE.json: nDCG@20 parity vs ref) on the 200-document ranking lists."""
    from capreolus_amd import run_io
    from tests.helpers import run_from_scores, synthetic_qrels

    c = load_case(kind, "ranklist")
    if kind == "knrm":
        mu, sigma, w1, b1, w2, b2 = knrm_weights(c)
        got, _ = oracle.knrm(c["query"], c["posdoc"], oracle.pack(c["emb"]), int(c["D"]), mu, sigma, w1, b1, w2, b2, bool(c["scoretanh"]))
    else:
        got, _, _ = _drmm_run(c)
    vals = []
    for seed in range(5):
        qrels = {"1": synthetic_qrels(len(got), seed)}
        ours = run_io.ndcg_cut(qrels, {"1": run_from_scores(got)}, 20)["1"]
        ref = run_io.ndcg_cut(qrels, {"1": run_from_scores(c["ref_scores"])}, 20)["1"]
        vals.append((ours, ref))
    if kind == "knrm":
        assert all(abs(a - b) < 1e-12 for a, b in vals), vals   # identical fp16 scores -> identical ranking
    else:
        # DRMM: the reference's own cos(a,a) < 1 coin flips move ~1% of some scores; the metric moves with them
        assert np.mean([abs(a - b) for a, b in vals]) < 0.05, vals


def _tks_oracle(c):
    return oracle.drmmtks(c["query"], c["posdoc"], c["query_idf"], oracle.pack(c["emb"]), int(c["D"]), int(c["topk"]), c["sd.gates.weight"],
                          c["sd.ffw.0.weight"], c["sd.ffw.0.bias"], c["sd.output_layer.weight"], c["sd.output_layer.bias"])


@pytest.mark.parametrize("name", ["default", "top3_short", "ranklist", "multiquery"])
def test_drmmtks_oracle_matches_reference(name):
    c = load_case("drmmtks", name)
    got, err = _tks_oracle(c)
    assert err == 0
    assert rel_err(got, c["ref_scores"]).max() <= REL_TOL, (name, rel_err(got, c["ref_scores"]).max())
    if name in ("ranklist", "multiquery"):
        assert (got.astype(np.float16) == c["ref_scores_f16"]).mean() > 0.98


@pytest.mark.parametrize("name", ["default", "tanh_noidf_short", "ranklist", "multiquery"])
def test_pacrr_oracle_matches_reference(name):
    from tests.helpers import pacrr_args

    c = load_case("pacrr", name)
    got, err = oracle.pacrr(c["query"], c["posdoc"], c["query_idf"], oracle.pack(c["emb"]), int(c["D"]), *pacrr_args(c))
    assert err == 0
    assert rel_err(got, c["ref_scores"]).max() <= REL_TOL, (name, rel_err(got, c["ref_scores"]).max())
    if name in ("ranklist", "multiquery"):
        assert (got.astype(np.float16) == c["ref_scores_f16"]).mean() > 0.98


@pytest.mark.parametrize("name", ["default", "nocross_2fc_short", "ranklist"])
def test_convknrm_oracle_matches_reference(name):
    from tests.helpers import convknrm_args

    c = load_case("convknrm", name)
    got, err = oracle.convknrm(c["query"], c["posdoc"], c["emb"], *convknrm_args(c))
    assert err == 0
    assert rel_err(got, c["ref_scores"]).max() <= REL_TOL, (name, rel_err(got, c["ref_scores"]).max())
    if name == "ranklist":
        assert (got.astype(np.float16) == c["ref_scores_f16"]).mean() > 0.98


@pytest.mark.parametrize("name", ["mini", "mini_max_single", "mini_nocls", "base"])
def test_cedr_port_matches_reference(name):
    import torch

    from oracle import bert_port
    from tests.helpers import load_cedr_case

    c, w, head, mus, sigmas = load_cedr_case(name)
    heads, layers = int(c["dims"][2]), int(c["dims"][1])
    t = [torch.from_numpy(c[k].astype(np.int64)) for k in ("pos_bert_input", "pos_mask", "pos_seg")]
    got = bert_port.cedr_knrm(w, head, *t, heads, layers, int(c["maxqlen"]), [int(x) for x in c["simmat_layers"]], mus, sigmas,
                              c["cls_mode"]).numpy()
    assert rel_err(got, c["ref_scores"]).max() <= 2e-5, (name, rel_err(got, c["ref_scores"]).max())
