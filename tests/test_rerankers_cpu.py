"""Host-side checks of the reranker mirrors that need no GPU: every mirror exposes exactly the parameter names the reference
module's state_dict has (read off the golden fixtures, which were generated from the reference modules), refuses CPU tensors
instead of falling back, and refuses training."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from capreolus_amd import reranker as rr
from tests.helpers import GOLDEN, load_case


def _sd_names(kind, name):
    z = np.load(os.path.join(GOLDEN, f"{kind}_{name}.npz"))
    return {k[3:] for k in z.files if k.startswith("sd.")}


def _batch(c):
    return {k: torch.as_tensor(c[k]) for k in ("query", "posdoc", "query_idf")}


def test_registry_lists_every_scored_model():
    assert sorted(rr.registry) == ["CEDRKNRM", "ConvKNRM", "DRMM", "DRMMTKS", "KNRM", "PACRR", "ptBERTMaxP"]
    for name, cls in rr.registry.items():
        assert cls.module_name == name


def test_pacrr_mirror_names_and_no_fallback():
    c = load_case("pacrr", "default")
    cfg = {k: int(c[f"cfg.{k}"]) for k in ("mingram", "maxgram", "nfilters", "kmax", "combine")}
    cfg.update(idf=bool(int(c["cfg.idf"])), nonlinearity=str(c["nonlinearity"]))
    r = rr.PACRR(cfg, SimpleNamespace(embeddings=c["emb"], config={"maxqlen": c["query"].shape[1]}))
    m = r.build_model()
    assert set(m.state_dict()) - {"embedding.weight"} == _sd_names("pacrr", "default")
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r.test(_batch(c))
    m.train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):      # training mode: the similarity matrix is still the HIP kernel's
        with torch.enable_grad():
            r.test(_batch(c))
    with pytest.raises(ValueError):
        rr.PACRR(dict(cfg, nonlinearity="gelu"), SimpleNamespace(embeddings=c["emb"], config={"maxqlen": 4})).build_model()


def test_convknrm_mirror_names_and_no_fallback():
    c = load_case("convknrm", "nocross_2fc_short")
    cfg = {k: int(c[f"cfg.{k}"]) for k in ("maxngram", "filters")}
    cfg.update({k: bool(int(c[f"cfg.{k}"])) for k in ("gradkernels", "crossmatch", "scoretanh", "singlefc")})
    r = rr.ConvKNRM(cfg, SimpleNamespace(embeddings=c["emb"], pad=0))
    m = r.build_model()
    convs = {f"convs.{g}.0.{p}" for g in range(cfg["maxngram"]) for p in ("weight", "bias")}   # (regenerated from a seed in the fixtures)
    assert set(m.state_dict()) - {"embeddings.weight"} - convs == _sd_names("convknrm", "nocross_2fc_short")
    assert convs <= set(m.state_dict())
    assert m.combine[0].in_features == 11 * cfg["maxngram"] and len(m.combine) == 4             # Linear, Tanh, Linear, Tanh
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r.test(_batch(c))


def test_cedrknrm_mirror_names():
    dims = dict(hidden=128, layers=2, heads=2, ffn=256, vocab=300, max_pos=64)
    cfg = dict(rr.CEDRKNRM.config_spec, pretrained=dims, simmat_layers=[0, 1, 2], combine_hidden=8)
    r = rr.CEDRKNRM(cfg, SimpleNamespace(config={"numpassages": 2, "maxseqlen": 32, "maxqlen": 6}))
    m = r.build_model()
    names = set(m.state_dict())
    assert {"one", "zero", "combine.0.weight", "combine.1.bias", "kernels.kernels.10.mu", "kernels.kernels.0.sigma",
            "bert.embeddings.word_embeddings.weight", "bert.embeddings.LayerNorm.bias", "bert.pooler.dense.weight",
            "bert.encoder.layer.1.attention.self.query.weight", "bert.encoder.layer.0.output.LayerNorm.weight"} <= names
    assert m.combine[0].in_features == 128 + 11 * 3 and m.maxqlen == 7
    assert float(m.kernels.kernels[10].mu.detach()) == 1.0 and abs(float(m.kernels.kernels[10].sigma.detach()) - 0.01) < 1e-9   # the exact-match kernel (:43-44)
    m.eval()
    x = torch.zeros((1, 2, 32), dtype=torch.int64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r.test({"pos_bert_input": x, "pos_mask": x, "pos_seg": x})
    # an ELECTRA-shaped body (the reference's default checkpoint family): the same encoder without a pooler
    e = rr.CEDRKNRM(dict(cfg, pretrained=dict(dims, pooler=False)), r.extractor).build_model()
    assert not any("pooler" in k for k in e.state_dict()) and "bert.encoder.layer.1.output.dense.bias" in e.state_dict()
    with pytest.raises(AssertionError):
        rr.CEDRKNRM(dict(cfg, simmat_layers=[-1], cls=None), r.extractor).build_model()


def test_fused_step_available_knows_the_per_batch_limits_before_the_optimizer_is_built():
    """`fused_step_available` answers from the configuration and the extractor's maxqlen what `fused_train_step` would only find out on
    the first batch (a `None` from the first batch leaves the trainer on eager steps instead of the captured-graph route)."""
    emb = np.zeros((10, 60), dtype=np.float32)
    ck = lambda maxqlen, **kw: rr.ConvKNRM(dict(rr.ConvKNRM.config_spec, **kw), SimpleNamespace(embeddings=emb, config={"maxqlen": maxqlen}))
    assert ck(4).fused_step_available(32) and ck(8).fused_step_available(512)
    assert not ck(9).fused_step_available(32)                  # 3 n-gram sizes x 9 query terms > 24 rows
    assert ck(24, crossmatch=False).fused_step_available(32)   # without crossmatch a view meets its own size only
    assert not ck(4).fused_step_available(513) and not ck(4, singlefc=False).fused_step_available(32) and not ck(4, filters=130).fused_step_available(32)
    dr = lambda maxqlen: rr.DRMM(dict(rr.DRMM.config_spec), SimpleNamespace(embeddings=emb, config={"maxqlen": maxqlen}))
    assert dr(4).fused_step_available(128) and not dr(4).fused_step_available(129) and not dr(8).fused_step_available(128) and dr(8).fused_step_available(64)
