"""Host logic on CPU: trainer mirror (batch filling, fp16 rounding, run writing), query sharding and the
world_size-2 gather over gloo, run IO and nDCG@20."""
import os
import socket
import sys
import zlib

import numpy as np
import pytest
import torch

from capreolus_amd import run_io
from capreolus_amd.trainer.pytorch import PytorchTrainer, shard_bounds, shard_pred_data


class FakeSampler(torch.utils.data.IterableDataset):
    """Stands in for the reference PredSampler (sampler/__init__.py:207-264): same attributes the trainer uses."""

    def __init__(self, n_queries=7, seed=0):
        rs = np.random.RandomState(seed)
        self.qid_to_docids = {str(300 + q): [f"d{q}_{i}" for i in range(rs.randint(1, 40))] for q in range(n_queries)}

    def _vec(self, qid, docid):
        h = (zlib.crc32(f"{qid}/{docid}".encode()) % 1000) / 7.0
        return {"qid": qid, "posdocid": docid, "query": np.full(4, int(qid), dtype=np.int64),
                "posdoc": np.full(16, len(docid), dtype=np.int64), "query_idf": np.full(4, h, dtype=np.float32)}

    def __iter__(self):
        for qid, docids in self.qid_to_docids.items():
            for d in docids:
                yield self._vec(qid, d)

    def __len__(self):
        return sum(len(v) for v in self.qid_to_docids.values())

    def get_qid_docid_pairs(self):
        for qid, docids in self.qid_to_docids.items():
            for d in docids:
                yield qid, d


class FakeReranker:
    """A scorer that is a pure function of the batch tensors (no HIP needed): exercises only the host logic."""

    def __init__(self):
        self.model = torch.nn.Linear(1, 1)
        self.calls = []

    def test(self, d):
        self.calls.append(len(d["qid"]))
        return d["query_idf"][:, 0] * 1.0009765625 + d["posdoc"][:, 0].float()


def _expected(s):
    out = {}
    for qid, docid in s.get_qid_docid_pairs():
        v = s._vec(qid, docid)
        sc = np.float32(v["query_idf"][0]) * np.float32(1.0009765625) + np.float32(v["posdoc"][0])
        out.setdefault(qid, {})[docid] = np.float32(sc).astype(np.float16).item()
    return out


def test_predict_single_process(tmp_path):
    s, r = FakeSampler(), FakeReranker()
    t = PytorchTrainer({"batch": 8, "coalesce": 0})       # the reference's control flow: one scoring call per DataLoader batch
    preds = t.predict(r, s, tmp_path / "sub" / "run.txt")
    assert preds == _expected(s)
    assert set(r.calls) == {8}  # the short last batch is filled by repetition (reference :339-340)
    # default: DataLoader batches are coalesced into few large scoring calls - same predictions, far fewer launches
    r2 = FakeReranker()
    assert PytorchTrainer({"batch": 8}).predict(r2, s) == preds and r2.calls == [len(s)]
    r3 = FakeReranker()
    assert PytorchTrainer({"batch": 8, "coalesce": 20}).predict(r3, s) == preds
    assert sum(r3.calls) == len(s) and all(c == 24 for c in r3.calls[:-1]) and len(r3.calls) == -(-len(s) // 24)
    run = run_io.load_trec_run(tmp_path / "sub" / "run.txt")
    assert list(run.keys()) == sorted(preds.keys(), key=int)
    for qid in run:
        sc = list(run[qid].values())
        assert sc == sorted(sc, reverse=True)
        assert run[qid] == pytest.approx(preds[qid])


def test_fill_incomplete_batch():
    t = PytorchTrainer({"batch": 5})
    b = {"qid": ["1", "2"], "x": torch.arange(4).view(2, 2), "y": np.arange(2)}
    f = t.fill_incomplete_batch(b)
    assert f["qid"] == ["1", "2", "1", "1", "1"]
    assert f["x"].tolist() == [[0, 1], [2, 3], [0, 1], [2, 3], [0, 1]]
    assert f["y"].tolist() == [0, 0, 0, 1, 1]  # numpy repeat is element-wise: the reference quirk is kept


def test_bad_config():
    with pytest.raises(ValueError):
        PytorchTrainer({"batch": 0})
    with pytest.raises(ValueError):
        PytorchTrainer({"amp": "yes"})
    with pytest.raises(ValueError):
        PytorchTrainer({"nonsense": 1})


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_bounds_cover_everything(world):
    rs = np.random.RandomState(world)
    for _ in range(20):
        sizes = rs.randint(1, 1000, size=rs.randint(1, 30)).tolist()
        b = shard_bounds(sizes, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == len(sizes)
        assert all(b[i] <= b[i + 1] for i in range(world))
    s = FakeSampler(11)
    seen = []
    for r in range(world):
        part, off, cnt, tot = shard_pred_data(s, r, world)
        assert tot == len(s) and off == len(seen)
        got = [(v["qid"], v["posdocid"]) for v in part]
        assert len(got) == cnt
        seen.extend(got)
    assert seen == list(s.get_qid_docid_pairs())


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, r = FakeSampler(9, seed=3), FakeReranker()
    preds = PytorchTrainer({"batch": 4}).predict(r, s, os.path.join(out_dir, "run.txt"))
    ok = preds == _expected(s)
    # each rank scored only its own block of queries
    n_local = shard_pred_data(s, rank, world)[2]
    ok = ok and sum(r.calls) >= n_local and sum(r.calls) < n_local + 4
    open(os.path.join(out_dir, f"ok{rank}"), "w").write(str(ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_predict_sharded_gloo(tmp_path, world):
    import torch.multiprocessing as mp

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"ok{r}").read() == "True"
    run = run_io.load_trec_run(tmp_path / "run.txt")
    assert sum(len(v) for v in run.values()) == len(FakeSampler(9, seed=3))


def test_ndcg_cut_hand_computed():
    qrels = {"1": {"a": 2, "b": 0, "c": 1, "d": 1}, "2": {"x": 1}}
    run = {"1": {"a": 0.5, "b": 0.9, "c": 0.5, "zz": 0.7, "d": 0.1}, "2": {"y": 1.0}, "3": {"q": 1.0}}
    # query 1 ranking: b(0) zz(unjudged 0) then the 0.5 tie broken by docid descending: c(1) a(2), then d(1)
    dcg = 0 / np.log2(2) + 0 / np.log2(3) + 1 / np.log2(4) + 2 / np.log2(5) + 1 / np.log2(6)
    idcg = 2 / np.log2(2) + 1 / np.log2(3) + 1 / np.log2(4)
    got = run_io.ndcg_cut(qrels, run, k=20)
    assert got["1"] == pytest.approx(dcg / idcg, abs=1e-12)
    assert got["2"] == 0.0 and "3" not in got
    assert run_io.ndcg_cut(qrels, run, k=2)["1"] == 0.0
    assert run_io.mean_ndcg_cut(qrels, run) == pytest.approx((dcg / idcg) / 2)


def test_ndcg_cut_published_vectors():
    """run_io.ndcg_cut (the host twin of capamd_ndcg_cut and the dev metric of PytorchTrainer.train) against numbers printed by
    pytrec_eval's README and the worked example of the DCG article (tests/helpers.py: NDCG_PUBLISHED)."""
    from tests.helpers import NDCG_PUBLISHED

    for source, qrels, run, k, want in NDCG_PUBLISHED:
        got = run_io.ndcg_cut(qrels, run, k)
        assert set(got) == set(want), source
        for qid, (v, tol) in want.items():
            assert abs(got[qid] - v) <= tol, (source, qid, got[qid], v)
    # trec_eval breaks score ties by docid DESCENDING, whatever order the run lists them in (its README: "ties are broken
    # deterministically (using docno)"; form_res_rels sorts by sim, then docno, both descending)
    qrels = {"1": {"a": 1, "b": 0}}
    for run in ({"1": {"a": 1.0, "b": 1.0}}, {"1": {"b": 1.0, "a": 1.0}}):
        assert run_io.ndcg_cut(qrels, run, 1)["1"] == 0.0          # "b" > "a": b is ranked first
        assert run_io.ndcg_cut(qrels, run, 2)["1"] == pytest.approx(1 / np.log2(3))


def test_abi_header_and_library_agree():
    """Every function include/capreolus_amd.h declares is exported by the built library, and the
    ctypes table binds exactly that set (no compute calls: there is no GPU here)."""
    import re

    from capreolus_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "capreolus_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(capamd_\w+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.capamd_version() == int(re.search(r"#define CAPAMD_VERSION (\d+)", hdr).group(1))
    assert lib.capamd_arch() == b"gfx950"
    assert lib.capamd_packed_row_stride(300) == 320 and lib.capamd_packed_row_stride(63) == 64
    assert lib.capamd_packed_row_stride(64) == 128 and lib.capamd_packed_row_stride(320) == -1
    assert lib.capamd_packed_table_bytes(400001, 300) == 400001 * 320 * 4


def test_no_compiler_generated_read_sits_too_close_behind_an_inline_mfma():
    """csrc/hazard_lint.py: the hazard LLVM cannot see (an accumulator spilled or copied right behind an inline-assembly MFMA) is
    reported on a listing that has it, not on one that waits, and is absent from every object the libraries are linked from."""
    import glob
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from capreolus_amd.csrc import build as hipbuild

    hazard_lint = hipbuild._hazard_lint()

    mfma = "\tv_mfma_f32_16x16x32_bf16 a[208:211], v[80:83], v[96:99], a[208:211]"
    early = ["k:", mfma, "\ts_nop 0", "\tv_accvgpr_read_b32 v100, a208"]
    found = hazard_lint.lint_listing(early)
    assert len(found) == 1 and found[0][3] == 1 and found[0][4] == 6
    assert hazard_lint.lint_listing(["k:", mfma, "\ts_nop 5", "\tv_accvgpr_read_b32 v100, a208"]) == []
    assert hazard_lint.lint_listing(["k:", mfma, "\tv_accvgpr_read_b32 v100, a212"]) == []          # another tile's register
    assert hazard_lint.lint_listing(["k:", mfma, mfma.replace("v[80:83]", "v[84:87]")]) == []          # MFMA on MFMA: interlocked
    assert hazard_lint.lint_listing(["k:", mfma, "\ts_branch 12", "\tv_accvgpr_read_b32 v100, a208"]) == []
    wide = "\tv_mfma_f32_32x32x16_f16 v[0:15], v[20:23], v[24:27], v[0:15]"
    assert len(hazard_lint.lint_listing(["k:", wide] + ["\ts_nop 0"] * 9 + ["\tv_mul_f32_e32 v40, v41, v7"])) == 1
    assert hazard_lint.lint_listing(["k:", wide] + ["\ts_nop 0"] * 10 + ["\tv_mul_f32_e32 v40, v41, v7"]) == []
    # edges, not only text: a read at the TARGET of a branch behind the MFMA (forward: over the waiting path; backward: a loop header
    # reached from the loop's tail) is seen; the same read far enough down either path is not
    fwd = ["k:", mfma, "\ts_cbranch_scc1 .LBB0_2", "\ts_nop 7", "\ts_branch .LBB0_3", ".LBB0_2:", "\tv_accvgpr_read_b32 v100, a208", ".LBB0_3:", "\ts_endpgm"]
    found = hazard_lint.lint_listing(fwd)
    assert len(found) == 1 and "v_accvgpr_read_b32" in found[0][2]
    assert hazard_lint.lint_listing([l.replace("v100, a208", "v100, a212") for l in fwd]) == []
    back = ["k:", ".LBB0_1:", "\tv_accvgpr_read_b32 v100, a208", "\ts_nop 7", mfma, "\ts_cbranch_scc1 .LBB0_1", "\ts_nop 7", "\ts_endpgm"]
    found = hazard_lint.lint_listing(back)
    assert len(found) == 1 and found[0][3] == 1 + hazard_lint.kTakenBranchStates
    assert hazard_lint.lint_listing(["k:", ".LBB0_1:", "\ts_nop 7", "\tv_accvgpr_read_b32 v100, a208", mfma, "\ts_cbranch_scc1 .LBB0_1", "\ts_endpgm"]) == []
    # llvm-objdump's form: addresses in the trailing comment, the target as <function+0xoffset>
    dis = ["0000000000001000 <k>:", mfma + "   // 000000001000: D3B50000", "\ts_cbranch_scc1 2   // 000000001008: BF850002 <k+0x14>",
           "\ts_nop 7   // 00000000100C: BF800007", "\ts_endpgm   // 000000001010: BF810000", "\tv_accvgpr_read_b32 v100, a208   // 000000001014: D3D84064",
           "\ts_endpgm   // 00000000101C: BF810000"]
    assert len(hazard_lint.lint_listing(dis)) == 1
    if hazard_lint.objdump() is None:
        pytest.skip("no llvm-objdump in this toolchain: the listing checks above ran, the objects cannot be disassembled")
    objs = glob.glob(os.path.join(root, "capreolus_amd", "csrc", "*.o"))
    assert len(objs) >= 17
    for obj in objs:
        assert hazard_lint.lint_object(obj) == [], obj


def test_abi_entries_reject_null_pointers():
    """Error behaviour at the boundary: every status-returning entry answers CAPAMD_ERR_ARG to null pointers before it touches the
    device (argument checks come first, so this runs without a GPU) - a caller's mistake is an error code, never a crash."""
    import ctypes

    from capreolus_amd import _lib

    lib = _lib.load()

    def is_ptr(t):
        return t is ctypes.c_void_p or hasattr(t, "contents")

    checked = 0
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        if restype is not ctypes.c_int or not any(is_ptr(t) for t in argtypes):
            continue
        args = [None if is_ptr(t) else 1 for t in argtypes]
        assert getattr(lib, name)(*args) == _lib.ERR_ARG, name
        checked += 1
    assert checked >= 20

    # geometry limits are refused the same way (device pointers are never dereferenced on the host: any aligned non-null value will do)
    P = 0x10000
    knrm = lambda Q=4, L=800, D=300, K=11, hidden=0, V=1000: lib.capamd_knrm_forward(P, P, 8, Q, L, P, V, D, P, P, K, P, P, hidden, P, P, 0, P, P, None, 0, 0, None)
    # ({"L": 24000}: the term list of such a document does not fit the 160 KiB of LDS a workgroup can have - refused, not mis-launched)
    for kw in ({"D": 320}, {"K": 13}, {"K": 0}, {"Q": 0}, {"L": 0}, {"L": 32769}, {"L": 24000}, {"hidden": -1}, {"hidden": 10**6}, {"V": 0}, {"V": 2**31}):
        assert knrm(**kw) == _lib.ERR_ARG, kw
    pacrr = lambda Q=4, L=800, maxgram=3, kmax=2, nf=32, comb=32: lib.capamd_pacrr_forward(
        P, P, P, 8, Q, L, P, 1000, 300, 1, maxgram, nf, kmax, P, P, 1, comb, 1, P, P, P, P, P, P, P, P, None)
    for kw in ({"Q": 9}, {"L": 1025}, {"maxgram": 4}, {"kmax": 5}, {"nf": 257}, {"comb": 129}):
        assert pacrr(**kw) == _lib.ERR_ARG, kw
    assert lib.capamd_convknrm_table_bytes(1000, 4, 128) == -1 and lib.capamd_convknrm_table_bytes(1000, 3, 24) == -1
    assert lib.capamd_convknrm_table_bytes(1000, 3, 128) == 1000 * 6 * 128 * 4
    assert lib.capamd_pack_embeddings(P, 10, 300, 300, P + 16, None) == _lib.ERR_ALIGN       # packed rows are 256-byte aligned


# ---- training control flow on the host (reference trainer/pytorch.py:47-122, 124-300; trainer/__init__.py:98-109) ----

def test_build_checks_and_lr_schedule():
    for bad in ({"niters": 0}, {"niters": 2, "validatefreq": 3}, {"itersize": 8, "batch": 16}, {"gradacc": 0}, {"gradacc": 1.5}, {"lr": 0.0},
                {"decaytype": "cosine"}):
        with pytest.raises(ValueError):
            PytorchTrainer(bad)
    t = PytorchTrainer({"batch": 4, "itersize": 16, "warmupiters": 2, "decaytype": "linear", "decay": 0.5})
    assert t.n_batch_per_iter == 4
    # warm-up over 2 iterations x 4 steps, then 1 / (1 + decay * epochs since the warm-up)
    assert [t.lr_multiplier(s) for s in (0, 3, 7, 8)] == [1 / 8, 4 / 8, 1.0, 1.0]
    assert t.lr_multiplier(12) == pytest.approx(1 / 1.5) and t.lr_multiplier(16) == pytest.approx(1 / 2.0)
    e = PytorchTrainer({"batch": 4, "itersize": 16, "decaytype": "exponential", "decay": 0.1, "decayiters": 2})
    assert e.lr_multiplier(0) == 1.0 and e.lr_multiplier(8) == pytest.approx(0.1) and e.lr_multiplier(4) == pytest.approx(0.1 ** 0.5)
    assert PytorchTrainer({}).lr_multiplier(1000) == 1
    # seeding happens in build(), as in the reference (:73-74)
    PytorchTrainer({"seed": 7})
    a = torch.rand(3)
    PytorchTrainer({"seed": 7})
    assert torch.equal(a, torch.rand(3))


class _PairData(torch.utils.data.IterableDataset):
    def __init__(self):
        self.served = 0

    def __iter__(self):
        g = torch.Generator().manual_seed(0)
        while True:
            self.served += 1
            x = torch.randn(3, generator=g)
            yield {"pos": x + 1.0, "neg": x - 1.0}


class _LinearReranker:
    """score() on CPU tensors, the reference's checkpoint format through the Reranker base class."""

    def __init__(self):
        from capreolus_amd.reranker import Reranker

        torch.manual_seed(0)
        self.model = torch.nn.Linear(3, 1)
        self._base = Reranker.__new__(Reranker)
        self._base.model = self.model

    def score(self, d):
        return [self.model(d["pos"]).view(-1), self.model(d["neg"]).view(-1)]

    def test(self, d):
        return d["query_idf"][:, 0]

    def save_weights(self, fn, opt):
        self._base.save_weights(fn, opt)

    def load_weights(self, fn, opt):
        self._base.load_weights(fn, opt)


def test_train_loop_schedule_gradacc_and_fastforward(tmp_path):
    s = FakeSampler(3)
    qrels = {q: {d: (i % 3) for i, d in enumerate(ds)} for q, ds in s.qid_to_docids.items()}
    cfg = {"batch": 4, "itersize": 16, "niters": 3, "lr": 0.01, "gradacc": 2, "warmupiters": 1, "fastforward": True, "evalbatch": 8}
    t, r = PytorchTrainer(cfg), _LinearReranker()
    lrs = []
    step = torch.optim.Adam.step

    def spy(self, *a, **k):
        lrs.append(self.param_groups[0]["lr"])
        return step(self, *a, **k)

    torch.optim.Adam.step = spy
    try:
        data = _PairData()
        losses = t.train(r, data, tmp_path / "train", s, tmp_path / "dev", qrels, "ndcg_cut_20", relevance_level=2)
    finally:
        torch.optim.Adam.step = step
    assert len(losses) == 3 and losses[-1] < losses[0]
    # 4 batches per iteration, an update every 2nd; the schedule is advanced after every batch but the last of an iteration with
    # step = iteration * 4 + batch (reference :88, 118-120): update k of iteration i sees multiplier(i*4 + 2k - 1 ... ) as below
    mult = [t.lr_multiplier(x) for x in (4, 6, 8, 10, 12, 14)]
    assert lrs == pytest.approx([0.01 * m for m in mult])
    assert data.served == 3 * 4 * 4 + 1 or data.served == 3 * 4 * 4        # 12 batches of 4 (the loader may prefetch one sample)
    info = (tmp_path / "train" / "info" / "loss.txt").read_text().splitlines()
    assert [ln.split()[0] for ln in info] == ["0", "1", "2"]
    assert (tmp_path / "train" / "dev.best").exists() and (tmp_path / "train" / "weights" / "3.p").exists()
    assert (tmp_path / "dev" / "metrics.json").exists() and (tmp_path / "dev" / "3.run").exists()
    # the dev metric is nDCG over the GRADED judgments whatever relevance_level is (pytrec_eval's ndcg_cut ignores the level)
    import json

    m = json.loads((tmp_path / "dev" / "metrics.json").read_text())["ndcg_cut_20"]
    runs = [run_io.mean_ndcg_cut(qrels, run_io.load_trec_run(tmp_path / "dev" / f"{i}.run"), 20) for i in (1, 2, 3)]
    assert m == pytest.approx(max(runs), abs=1e-6)
    # resume: 3 iterations are on disk -> the reference loads weights/<len(loss) - 1>.p and continues with iteration len(loss) + 1
    w3 = pickle_load(tmp_path / "train" / "weights" / "2.p")
    t2, r2 = PytorchTrainer(dict(cfg, niters=4)), _LinearReranker()
    losses2 = t2.train(r2, _PairData(), tmp_path / "train", s, tmp_path / "dev", qrels, "ndcg_cut_20")
    assert len(losses2) == 4 and losses2[:3] == pytest.approx(losses)
    assert (tmp_path / "train" / "weights" / "4.p").exists()
    assert set(w3) == {"weight", "bias"}


def pickle_load(fn):
    import pickle

    with open(fn, "rb") as f:
        return pickle.load(f)


@pytest.mark.parametrize("mode,exact,expect_lists", [("exact", True, True), ("exact", False, False), ("always", False, True), ("never", True, False)])
def test_resident_scoring_picks_lists_by_option(mode, exact, expect_lists):
    """`PytorchTrainer._score_store` (predict / predict_resident / evaluate_resident on a candidate store): whole candidate lists for the
    rerankers the `lists` option admits - "exact": only those whose list scores equal their per-pair scores bit for bit - with the
    lists' offsets handed over as a host array; the per-pair route in `evalbatch` steps otherwise, for more than four query terms and -
    bit-identical rerankers only - for runs of fewer than eight candidates per query and for a single list."""
    calls = []

    class Store:
        device = torch.device("cpu")
        q_table = torch.zeros(3, 4, dtype=torch.int32)
        d_table = torch.zeros(40, 800, dtype=torch.int32)

    class Fake:
        supports_lists = True
        lists_bit_identical = exact

        def test_resident_lists(self, store, pq, pd, offsets):
            calls.append(("lists", np.asarray(offsets).tolist()))
            return torch.arange(pq.numel(), dtype=torch.float32)

        def test_resident(self, store, pq, pd):
            calls.append(("pairs", int(pq.numel())))
            return pd.float()

    counts = [20, 9, 11]
    n = sum(counts)
    pq = torch.repeat_interleave(torch.arange(3, dtype=torch.int32), torch.tensor(counts))
    pd = torch.arange(n, dtype=torch.int32)
    tr = PytorchTrainer({"lists": mode})
    out = tr._score_store(Fake(), Store(), pq, pd, counts, 16)
    assert out.dtype == torch.float32 and out.numel() == n and torch.equal(out, torch.arange(n, dtype=torch.float32))
    if expect_lists:
        assert calls == [("lists", [0, 20, 29, 40])]
        # a run whose per-pair workspace (4 L + 32 bytes per pair) would pass the bound goes in calls of whole lists, same scores
        calls.clear()
        tr.LISTS_PAIR_BYTES = 30 * (4 * 800 + 32)
        parts = tr._score_store(Fake(), Store(), pq, pd, counts, 16)
        assert calls == [("lists", [0, 20, 29]), ("lists", [0, 11])] and parts.numel() == n
        del tr.LISTS_PAIR_BYTES
    else:
        assert calls == [("pairs", 16), ("pairs", 16), ("pairs", 8)]
    # more than four query terms, or short lists: always pair by pair
    calls.clear()
    wide = Store()
    wide.q_table = torch.zeros(3, 6, dtype=torch.int32)
    tr._score_store(Fake(), wide, pq, pd, counts, 64)
    assert calls == [("pairs", n)]
    # short lists and a single list: a reranker whose two routes give the same bits takes the per-pair kernels (they fill the chip
    # better); one whose list scores round differently (KNRM) keeps the route its configuration names WHATEVER the call holds - a rank's
    # shard of one query must give the bits the unsharded run gives (test_knrm_predictions_do_not_depend_on_the_sharding on the GPU)
    sticky = expect_lists and not exact
    calls.clear()
    tr._score_store(Fake(), Store(), pq[:6], pd[:6], [2, 2, 2], 64)
    assert calls == ([("lists", [0, 2, 4, 6])] if sticky else [("pairs", 6)])
    calls.clear()
    tr._score_store(Fake(), Store(), pq[:20], pd[:20], [20], 64)
    assert calls == ([("lists", [0, 20])] if sticky else [("pairs", 20)])
    calls.clear()
    tr._score_store(Fake(), Store(), pq[:1], pd[:1], [1], 64)
    assert calls == ([("lists", [0, 1])] if sticky else [("pairs", 1)])
    with pytest.raises(ValueError):
        PytorchTrainer({"lists": "sometimes"}).build()


def test_pyhost_builds_the_dictionaries_of_the_python_expression():
    """csrc/pyhost.c (the CPython helper `predict` builds its {qid: {docid: score}} result with) against the expression it replaces:
    same keys in the same insertion order, values float(np.float16(x)) - incl. inf / nan / -0.0 / subnormals -, later duplicates win,
    merge = dict.update semantics for a qid that comes in several runs, errors as Python exceptions."""
    from capreolus_amd import pyhost
    from capreolus_amd.csrc import build as hipbuild

    hipbuild.build_pyhost()
    assert pyhost.available()
    rs = np.random.RandomState(5)
    bits = rs.randint(0, 65536, size=5000).astype(np.uint16)      # every kind of fp16 value
    scores = bits.view(np.float16)
    groups, lo = [], 0
    for q in range(40):
        n = int(rs.randint(0, 200))
        docids = tuple(f"d{rs.randint(0, 150)}" for _ in range(n))      # duplicates inside a list
        groups.append((f"q{q % 25}" if q >= 30 else q, docids, lo))      # int and str qids; q30.. repeat qids (merge)
        lo += n
    groups = [g for g in groups if g[2] + len(g[1]) <= scores.size]

    def python_form(gs, merge):
        out = {}
        for qid, docids, at in gs:
            vals = scores[at:at + len(docids)].tolist()
            if merge and qid in out:
                out[qid].update(zip(docids, vals))
            else:
                out[qid] = dict(zip(docids, vals))
        return out

    def same(a, b):
        assert list(a) == list(b)
        for k in a:
            assert list(a[k]) == list(b[k])
            for d in a[k]:
                x, y = a[k][d], b[k][d]
                assert type(x) is float and type(y) is float
                assert (x != x and y != y) or (x == y and math.copysign(1, x) == math.copysign(1, y)), (k, d, x, y)

    import math

    for merge in (False, True):
        same(pyhost.preds_from_fp16(groups, scores, {}, merge=merge), python_form(groups, merge))
    # a slice of the groups against a slice of the score vector (how `predict` converts part after part)
    g0, g1 = 5, 17
    base = groups[g0][2]
    end = groups[g1 - 1][2] + len(groups[g1 - 1][1])
    part = np.ascontiguousarray(scores[base:end])
    same(pyhost.preds_from_fp16(groups, part, {}, g0, g1, base=base), python_form(groups[g0:g1], False))
    with pytest.raises(IndexError):
        pyhost.preds_from_fp16(groups, part, {}, g0, g1 + 1, base=base)
    with pytest.raises(TypeError):
        pyhost.preds_from_fp16(groups, scores.astype(np.float32), {})
    with pytest.raises(TypeError):
        pyhost.preds_from_fp16([("q", ["not", "a", "tuple"], 0)], scores, {})
