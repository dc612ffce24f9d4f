"""BERT-MaxP on the GPU: unit checks of the MFMA GEMM and the fused attention against plain fp32 PyTorch on
the same bf16-rounded operands, and end-to-end parity with the fp32 oracle / the reference golden vectors."""
import ctypes
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from capreolus_amd import _lib, engine
from capreolus_amd.reranker import PTBERTMaxP
from oracle import bert_port
from tests.helpers import BERT_CASES, load_bert_case, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# bf16 operands (8 mantissa bits) and a bf16 activation stream through 12 layers: observed 3e-3 .. 1.2e-2 relative
# on passage logits (BERT-base fixture: 1.13e-2); the north-star 1e-3 is met by the fp16 default below, not by a
# bf16 MFMA path (SURVEY.md §7 "BERT parity in bf16": gate on rank order + a documented looser tolerance).
BF16_E2E_TOL = 2e-2
# fp16 operands (11 significant bits), the engine default: observed <= 7e-4 -> the north-star 1e-3 holds
FP16_E2E_TOL = 1e-3
TDT = {"bf16": (torch.bfloat16, 0, 2 ** -7), "fp16": (torch.float16, 1, 2 ** -10)}


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 768), (256, 2304, 768), (256, 768, 3072), (64, 64, 64), (128, 192, 320)])
@pytest.mark.parametrize("epi", [0, 1, 4])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_gemm_vs_torch(M, N, K, epi, dt):
    tdt, code, rtol = TDT[dt]
    g = torch.Generator(device=DEV).manual_seed(M + N + K + epi)
    A = (torch.randn((M, K), generator=g, device=DEV) * 0.5).to(tdt)
    # asymmetric, non-random structure so that a transposed/misplaced tile cannot cancel out
    W = (torch.randn((N, K), generator=g, device=DEV) * 0.05 + torch.arange(N, device=DEV)[:, None] * 1e-3).to(tdt)
    bias = torch.randn(N, generator=g, device=DEV)
    resid = torch.randn((M, N), generator=g, device=DEV)
    if epi == 4:
        resid = resid.to(tdt)
    out = torch.empty((M, N), dtype=tdt, device=DEV)
    rc = _lib.load().capamd_bert_gemm(_p(A), _p(W), _p(bias), M, N, K, epi, _p(resid), _p(out), code, _stream())
    assert rc == 0
    ref = A.float() @ W.float().t() + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 4:
        ref = ref + resid.float()
    torch.testing.assert_close(out.float(), ref, rtol=rtol, atol=2e-2 if dt == "bf16" else 3e-3)  # one rounding of the result


def _to_cm(x):
    """row-major [M, C] -> the engine's chunk-major layout (include/capreolus_amd.h), flattened"""
    M, C = x.shape
    return x.reshape(M // 32, 32, C // 8, 8).permute(0, 2, 1, 3).contiguous().reshape(-1)


def _from_cm(flat, M, C):
    return flat.reshape(M // 32, C // 8, 32, 8).permute(0, 2, 1, 3).contiguous().reshape(M, C)


@pytest.mark.parametrize("M,N,K,epi", [(512, 768, 768, 0), (256, 3072, 768, 1), (768, 256, 256, 0)])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_gemm_folded_layernorm_consumer(M, N, K, epi, dt):
    """LN(P) W0^T + b computed from the un-normalised P with gamma-scaled weights and the (mu, rstd, cs, c) epilogue."""
    tdt, code, rtol = TDT[dt]
    g = torch.Generator(device=DEV).manual_seed(M + N + K + epi)
    P = (torch.randn((M, K), generator=g, device=DEV) * 1.7 + torch.randn((M, 1), generator=g, device=DEV) * 0.8).to(tdt)
    W0 = torch.randn((N, K), generator=g, device=DEV) * 0.04 + torch.arange(N, device=DEV)[:, None] * 2e-4
    b = torch.randn(N, generator=g, device=DEV) * 0.3
    gamma = 1.0 + 0.2 * torch.randn(K, generator=g, device=DEV)
    beta = 0.1 * torch.randn(K, generator=g, device=DEV)
    Pf = P.float()
    mu = Pf.mean(1)
    rstd = torch.rsqrt(Pf.var(1, unbiased=False) + 1e-12)
    Wg = W0 * gamma[None, :]
    Ws = (Wg - Wg.mean(1, keepdim=True)).to(tdt)  # centred rows, as the engine packs them: LN(P) sums to zero over k
    cs = Ws.float().sum(1)
    c = b + W0 @ beta
    mr = torch.stack([mu, rstd], 1).contiguous()
    out = torch.empty(M * N, dtype=tdt, device=DEV)
    P_cm = _to_cm(P)
    rc = _lib.load().capamd_bert_gemm_ln(_p(P_cm), _p(Ws), _p(c), M, N, K, epi | 0x300, _p(mu), _p(rstd), _p(mr), _p(cs), None, None, None, None,
                                         _p(out), code, _stream())
    assert rc == 0
    x = (Pf - mu[:, None]) * rstd[:, None] * gamma + beta
    ref = x @ W0.t() + b
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    torch.testing.assert_close(_from_cm(out, M, N).float(), ref, rtol=4 * rtol, atol=5e-2 if dt == "bf16" else 8e-3)


@pytest.mark.parametrize("M,N,K", [(512, 768, 768), (256, 768, 3072), (768, 256, 128)])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_gemm_residual_stats_producer(M, N, K, dt):
    """out = A W^T + b' + LN-without-beta(R), chunk-major, plus the row statistics of the rounded output."""
    tdt, code, rtol = TDT[dt]
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    A = (torch.randn((M, K), generator=g, device=DEV) * 0.5).to(tdt)
    W = (torch.randn((N, K), generator=g, device=DEV) * 0.05 + torch.arange(N, device=DEV)[:, None] * 1e-3).to(tdt)
    bp = torch.randn(N, generator=g, device=DEV)
    R = (torch.randn((M, N), generator=g, device=DEV) * 2.0 + 0.5).to(tdt)
    gamma = 1.0 + 0.2 * torch.randn(N, generator=g, device=DEV)
    Rf = R.float()
    mu = Rf.mean(1)
    rstd = torch.rsqrt(Rf.var(1, unbiased=False) + 1e-12)
    mr = torch.stack([mu, rstd], 1).contiguous()
    out = torch.empty(M * N, dtype=tdt, device=DEV)
    part = torch.zeros((M, N // 64, 2), device=DEV)
    R_cm = _to_cm(R)
    rc = _lib.load().capamd_bert_gemm_ln(_p(A), _p(W), _p(bp), M, N, K, 5 | 0x100, None, None, None, None, _p(R_cm), _p(mr), _p(gamma), _p(part),
                                         _p(out), code, _stream())
    assert rc == 0
    ref = A.float() @ W.float().t() + bp + (Rf - mu[:, None]) * rstd[:, None] * gamma
    got = _from_cm(out, M, N).float()
    torch.testing.assert_close(got, ref, rtol=rtol, atol=2e-2 if dt == "bf16" else 3e-3)
    # the statistics are those of the ROUNDED output, slice by slice
    want = torch.stack([got.reshape(M, N // 64, 64).sum(2), (got * got).reshape(M, N // 64, 64).sum(2)], 2)
    torch.testing.assert_close(part, want, rtol=1e-4, atol=1e-3)


def test_gemm_gelu_accuracy():
    """The fused erf-GELU on its own: x sweeps [-12, 12] through an identity weight; the result may differ from the exact
    erf form only by the fp16 rounding of the output plus the 1.2e-6 bound of the in-kernel approximation."""
    M, N, K = 256, 64, 64
    x = torch.linspace(-12, 12, M * K, device=DEV).reshape(M, K).to(torch.float16)
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-4, -1e-4, 65504.0, -65504.0, 3.6, -3.6], device=DEV, dtype=torch.float16)
    W = torch.eye(N, K, device=DEV, dtype=torch.float16)
    bias = torch.zeros(N, device=DEV)
    out = torch.empty((M, N), dtype=torch.float16, device=DEV)
    assert _lib.load().capamd_bert_gemm(_p(x), _p(W), _p(bias), M, N, K, 1, None, _p(out), 1, _stream()) == 0
    ref = torch.nn.functional.gelu(x.double())
    err = (out.double() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -11 + 1.5e-6).all()), float((err - ref.abs() * 2.0 ** -11).max())


@pytest.mark.parametrize("S,hidden,heads,npsg", [(32, 128, 2, 8), (64, 128, 2, 4), (96, 128, 2, 8), (128, 192, 3, 2), (160, 128, 2, 8), (192, 128, 2, 4),
                                                 (224, 128, 2, 8), (256, 768, 12, 2), (256, 768, 12, 3), (256, 128, 2, 5), (256, 1024, 16, 45), (384, 128, 2, 2), (512, 128, 2, 3)])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_qkv_attention_vs_torch(S, hidden, heads, npsg, dt):
    tdt, code, rtol = TDT[dt]
    atol = 2e-2 if dt == "bf16" else 3e-3
    g = torch.Generator(device=DEV).manual_seed(S + hidden)
    M = npsg * S
    x = torch.randn((M, hidden), generator=g, device=DEV).to(tdt)
    w = (torch.randn((3 * hidden, hidden), generator=g, device=DEV) * 0.06).to(tdt)
    b = torch.randn(3 * hidden, generator=g, device=DEV) * 0.1
    lens = torch.randint(min(5, S // 2), S + 1, (npsg,), generator=g, device=DEV)
    mask = (torch.arange(S, device=DEV)[None, :] < lens[:, None]).long()
    mask[0, 7] = 0  # a hole in the middle, not only a padded tail
    q, k, ctx = (torch.empty((M, hidden), dtype=tdt, device=DEV) for _ in range(3))
    vt = torch.empty((npsg * heads, 64, S), dtype=tdt, device=DEV)
    rc = _lib.load().capamd_bert_qkv_attention(_p(x), _p(w), _p(b), _p(mask), npsg, S, hidden, heads, _p(q), _p(k), _p(vt), _p(ctx), code, _stream())
    assert rc == 0
    qkv = x.float() @ w.float().t() + b
    qr, kr, vr = (t.view(npsg, S, heads, 64).transpose(1, 2) for t in qkv.split(hidden, dim=1))
    torch.testing.assert_close(q.float().view(npsg, S, heads, 64).transpose(1, 2), qr / 8, rtol=rtol, atol=atol)
    torch.testing.assert_close(k.float().view(npsg, S, heads, 64).transpose(1, 2), kr, rtol=rtol, atol=atol)
    torch.testing.assert_close(vt.float().view(npsg, heads, 64, S).transpose(2, 3), vr, rtol=rtol, atol=atol)
    # attention on the bf16 tensors the kernel itself consumed
    qb = q.float().view(npsg, S, heads, 64).transpose(1, 2)
    kb = k.float().view(npsg, S, heads, 64).transpose(1, 2)
    vb = vt.float().view(npsg, heads, 64, S).transpose(2, 3)
    att = torch.softmax(qb @ kb.transpose(-1, -2) + (1.0 - mask.float()).view(npsg, 1, 1, S) * torch.finfo(torch.float32).min, dim=-1)
    ref = (att @ vb).transpose(1, 2).reshape(M, hidden)
    torch.testing.assert_close(ctx.float(), ref, rtol=2e-2 if dt == "bf16" else 3e-3, atol=atol)  # P is rounded to 16 bits before PV


def _model(c, agg, dt="fp16"):
    pre = dict(hidden=c["hidden"], layers=c["layers"], heads=c["heads"], ffn=c["ffn"], vocab=c["vocab"], max_pos=c["max_pos"])
    B, P, S = c["pos_bert_input"].shape
    r = PTBERTMaxP({"pretrained": pre, "aggregation": agg, "compute_dtype": dt}, SimpleNamespace(config={"numpassages": P, "maxseqlen": S}))
    m = r.build_model()
    m.bert.load_state_dict(c["weights"], strict=True)
    m.to(DEV).eval()
    return r


@pytest.mark.parametrize("name", BERT_CASES)
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_bert_maxp_end_to_end(name, dt):
    """Against the reference's fp32 result.  What a 16-bit encoder can promise is an ABSOLUTE error on the passage logit (rounding of
    the GEMM operands through the layers: ~1e-2 in fp16 for classifier weights of this scale), so the relative figure depends on how
    large the logits happen to be: 7.5e-4 on `base` (logits ~ 11), 1.3e-2 of the batch's logit scale on `base_long` (|logit| < 1).
    The yardstick is the reference's OWN mixed-precision mode - amp = pred wraps `reranker.test` in autocast, trainer/pytorch.py:323-326 -
    whose passage logits the fixtures carry: this engine must not deviate more from fp32 than the reference's autocast does
    (x 1.25 in fp16, the default; x 1.5 in bf16)."""
    c = load_bert_case(name)
    d = {k: c[k].to(DEV) for k in ("pos_bert_input", "pos_mask", "pos_seg")}
    ref_l = c["ref_passage_logits"][:, 1]
    amp_err = np.abs(c["ref_passage_logits_amp_" + dt][:, 1] - ref_l).max()      # the reference's own autocast against its fp32 self
    bound = (1.25 if dt == "fp16" else 1.5) * amp_err + 1e-4      # (bf16: 8 mantissa bits, the draw-to-draw scatter of either side is larger)
    B, P, S = c["pos_bert_input"].shape
    for agg in ("max", "first", "sum", "avg"):
        r = _model(c, agg, dt)
        with torch.no_grad():
            got = r.test(d).cpu().numpy()
        err = np.abs(got - c["ref_" + agg]).max()
        assert err <= bound * (P if agg == "sum" else 1), (name, agg, dt, err, bound)
    # passage logits
    r = _model(c, "max", dt)
    with torch.no_grad():
        r.test(d)
        eng_out, plog = r.model._engine.forward(d["pos_bert_input"], d["pos_mask"], d["pos_seg"], "max", return_passage_logits=True)
    err = np.abs(plog.cpu().numpy() - ref_l).max()
    scale = np.abs(ref_l).max()
    print(f"{name} {dt}: max abs err on passage logits {err:.2e} (reference autocast {amp_err:.2e}); relative to the logit scale {err / scale:.2e}, "
          f"element-wise {rel_err(plog.cpu().numpy(), ref_l).max():.2e}")
    assert err <= bound, (name, dt, err, amp_err)
    # ... and, whatever the reference's autocast does on a fixture, a FIXED absolute bound on the passage logit per operand type
    # (observed: fp16 <= 1.2e-2, bf16 <= 1.1e-1 on the four fixtures): a regression in the attention / ring / folded-LayerNorm kernels
    # cannot hide behind a generous yardstick
    assert err <= (1.6e-2 if dt == "fp16" else 1.5e-1), (name, dt, err)
    # the north-star 1e-3 relative (fp16; 2e-2 with bf16 operands, 8 mantissa bits) on every fixture whose logits are of order 1 or more
    if scale >= 1.0:
        assert rel_err(plog.cpu().numpy(), ref_l).max() <= (FP16_E2E_TOL if dt == "fp16" else BF16_E2E_TOL), (name, dt)
    # rank order of the documents by MaxP score: wherever two reference scores are further apart than twice the error bound
    got_doc, ref_doc = eng_out.cpu().numpy(), c["ref_max"]
    for i in range(B):
        for j in range(B):
            if ref_doc[i] - ref_doc[j] > 2 * bound:
                assert got_doc[i] > got_doc[j], (name, dt, i, j)


@pytest.mark.parametrize("name", ["roberta_mini", "roberta_h256"])
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_roberta_maxp_end_to_end(name, dt):
    """A RoBERTa body behind PTBERTMaxP (reference ptBERTMaxP.py:46-48, 57-58; HF RobertaForSequenceClassification): pad-relative
    position ids, LayerNorm eps 1e-5, the dense -> tanh -> out_proj head, token types zeroed.  Same yardstick as the BERT fixtures."""
    from tests.helpers import load_roberta_case

    c = load_roberta_case(name)
    B, P, S = c["pos_bert_input"].shape
    pre = dict(arch="roberta", hidden=c["hidden"], layers=c["layers"], heads=c["heads"], ffn=c["ffn"], vocab=c["vocab"], max_pos=c["max_pos"])
    d = {k: c[k].to(DEV) for k in ("pos_bert_input", "pos_mask", "pos_seg")}
    ref_l = c["ref_passage_logits"][:, 1]
    amp_err = np.abs(c["ref_passage_logits_amp_" + dt][:, 1] - ref_l).max()
    bound = (1.25 if dt == "fp16" else 1.5) * amp_err + 1e-4
    for agg in ("max", "first", "sum", "avg"):
        r = PTBERTMaxP({"pretrained": pre, "aggregation": agg, "compute_dtype": dt}, SimpleNamespace(config={"numpassages": P, "maxseqlen": S}))
        m = r.build_model()
        m.bert.load_state_dict(c["weights"], strict=True)       # the reference's checkpoint names
        m.to(DEV).eval()
        with torch.no_grad():
            got = r.test(d).cpu().numpy()
        if agg == "avg":
            assert np.isnan(c["ref_avg"]).all() and np.isnan(got).all()        # 0 / 0 with zeroed token types, as the reference
        elif agg == "sum":
            assert (got == 0).all() and (c["ref_sum"] == 0).all()
        else:
            assert np.abs(got - c["ref_" + agg]).max() <= bound, (name, agg, dt, np.abs(got - c["ref_" + agg]).max(), bound)
    with torch.no_grad():
        for sp in (True, False):      # length buckets: pad-relative positions depend on the prefix only -> bit-identical
            out, plog = m._engine.forward(d["pos_bert_input"], d["pos_mask"], torch.zeros_like(d["pos_seg"]), "max", return_passage_logits=True, skip_padding=sp)
            err = np.abs(plog.cpu().numpy() - ref_l).max()
            assert err <= bound, (name, dt, sp, err, amp_err)
            if sp:
                first = plog.clone()
        assert torch.equal(first, plog)
    print(f"{name} {dt}: max abs err on passage logits {err:.2e} (reference autocast {amp_err:.2e})")
    with pytest.raises(NotImplementedError):
        PTBERTMaxP({"pretrained": "google/electra-base-discriminator"}, SimpleNamespace(config={"numpassages": P, "maxseqlen": S})).build_model()


@pytest.mark.parametrize("S,n_docs,P", [(64, 4, 4), (128, 4, 2), (192, 4, 1), (256, 3, 1), (256, 2, 3), (384, 2, 1), (512, 2, 2)])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_fused_layernorm_path_other_geometries(S, n_docs, P, dt):
    """hidden 256 / ffn 512 / 3 layers: every encoder GEMM is a ping-pong shape, so the folded-LayerNorm path runs (K = 256
    and 512: 4 and 8 K steps; N = 768, 256, 512) - at the three supported sequence lengths, against the fp32 ATen port."""
    from capreolus_amd.engine import BertEngine
    from oracle import bert_port

    heads, layers, vocab = 4, 3, 500
    w = bert_port.random_weights(hidden=256, layers=layers, heads=heads, ffn=512, vocab=vocab, max_pos=S, seed=S + P)
    g = torch.Generator().manual_seed(S * 7 + P)
    ids = torch.randint(1, vocab, (n_docs, P, S), generator=g)
    lens = torch.randint(8, S + 1, (n_docs, P), generator=g)
    mask = (torch.arange(S)[None, None, :] < lens[:, :, None]).long()
    seg = ((torch.arange(S)[None, None, :] >= 6) & (mask > 0)).long()
    ids = ids * mask
    if (n_docs * P * S) % 256:
        pytest.skip("rows of the microbatch must be a multiple of 256 for the fused path")
    ref = bert_port.maxp(w, ids, mask, seg, heads, layers, "max")
    eng = BertEngine({k: v.to(DEV) for k, v in w.items()}, heads, compute_dtype=dt)
    with torch.no_grad():
        got = eng.forward(ids.to(DEV), mask.to(DEV), seg.to(DEV), "max")
    # a 3-layer random model has small MaxP scores, so the RELATIVE error of a 16-bit encoder is larger here than on the
    # BERT-base fixture (measured: folded LayerNorm 1.2-1.8e-3, separate passes 2.4-2.9e-3 in fp16): path coverage, not the 1e-3 bar
    e = rel_err(got.cpu().numpy(), ref.numpy())
    assert e.max() <= (4e-2 if dt == "bf16" else 4e-3), (S, dt, e.max())


@pytest.mark.parametrize("name", BERT_CASES)
def test_length_buckets_give_identical_logits(name):
    """skip_padding: passages encoded at 64 / 128 / S by the position of their last attended token.  Pads get an attention
    weight of exactly 0 and never reach a real row, so every passage logit must come out bit-identical."""
    c = load_bert_case(name)
    d = {k: c[k].to(DEV).clone() for k in ("pos_bert_input", "pos_mask", "pos_seg")}
    B, P, S = d["pos_bert_input"].shape
    if S < 128:
        pytest.skip("nothing to bucket below S = 128")
    # make sure all buckets occur: cut some passages short, punch a hole into one mask
    g = torch.Generator().manual_seed(5)
    for b in range(B):
        for p in range(P):
            r = float(torch.rand(1, generator=g))
            if r < 0.35:
                n = int(torch.randint(6, 60, (1,), generator=g))
            elif r < 0.55 and S > 128:
                n = int(torch.randint(70, 128, (1,), generator=g))
            elif r < 0.75 and S > 192:
                n = int(torch.randint(130, 192, (1,), generator=g))
            else:
                continue
            d["pos_mask"][b, p, n:] = 0
            d["pos_bert_input"][b, p, n:] = 0
    d["pos_mask"][0, 0, 3] = 0
    for agg in ("max", "avg", "first"):
        r = _model(c, agg)
        eng_args = (d["pos_bert_input"], d["pos_mask"], d["pos_seg"], agg)
        if agg == "first":   # only passage 0 of every document is encoded at all
            with torch.no_grad():
                r.test(d)
                full = r.model._engine.forward(*eng_args, skip_padding=False)
                assert torch.equal(full, r.model._engine.forward(*eng_args, skip_padding=True))
            continue
        with torch.no_grad():
            r.test(d)
            full, pl_full = r.model._engine.forward(*eng_args, return_passage_logits=True, skip_padding=False)
            buck, pl_buck = r.model._engine.forward(*eng_args, return_passage_logits=True, skip_padding=True)
        assert torch.equal(pl_full, pl_buck)
        assert torch.equal(full, buck)


def test_bert_microbatching_and_errors():
    c = load_bert_case("mini")
    d = {k: c[k].to(DEV) for k in ("pos_bert_input", "pos_mask", "pos_seg")}
    r = _model(c, "max")
    with torch.no_grad():
        a = r.test(d).clone()
        r.model._engine.microbatch = 4  # 15 passages -> 4 micro-batches, last one short
        b = r.test(d)
        assert torch.equal(a, b)
        bad = {k: v.clone() for k, v in d.items()}
        bad["pos_bert_input"][0, 0, 3] = c["vocab"] + 5
        with pytest.raises(IndexError):
            r.test(bad)
        r.model.bert.classifier.bias.add_(0.5)  # live weights: no stale cache
        assert torch.allclose(r.test(d), a + 0.5, atol=1e-5)
    r.model.train()
    with pytest.raises(NotImplementedError):
        r.test(d)


# ---- CEDR-KNRM (row N4): encoder + per-layer masked cosine matrices + kernel pooling + combine ----

def _cedr_model(c, w, head, dt, skip_padding=True):
    from capreolus_amd.reranker import CEDRKNRM
    from tests.helpers import CEDR_MUS

    hidden, layers, heads, ffn, vocab, max_pos = (int(x) for x in c["dims"])
    P, S = c["pos_bert_input"].shape[1:]
    cfg = {"pretrained": dict(hidden=hidden, layers=layers, heads=heads, ffn=ffn, vocab=vocab, max_pos=max_pos), "mus": CEDR_MUS, "sigma": 0.1,
           "gradkernels": True, "hidden_dropout_prob": 0.1, "simmat_layers": [int(x) for x in c["simmat_layers"]],
           "combine_hidden": int(c["combine_hidden"]), "cls": c["cls_mode"], "compute_dtype": dt, "skip_padding": skip_padding}
    r = CEDRKNRM(cfg, SimpleNamespace(config={"numpassages": P, "maxseqlen": S, "maxqlen": int(c["maxqlen"])}))
    m = r.build_model()
    sd = dict(w)
    sd.pop("classifier.weight"), sd.pop("classifier.bias")
    sd.update(head)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("kernels.") or k in ("one", "zero") for k in missing), (missing, unexpected)   # the reference's names
    m.to(DEV).eval()
    return r


@pytest.mark.parametrize("name", ["mini", "mini_max_single", "mini_nocls", "base"])
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("skip_padding", [True, False])
def test_cedr_knrm_end_to_end(name, dt, skip_padding):
    from oracle import bert_port
    from tests.helpers import load_cedr_case

    c, w, head, mus, sigmas = load_cedr_case(name)
    r = _cedr_model(c, w, head, dt, skip_padding)
    d = {k: torch.from_numpy(c[k].astype(np.int64)).to(DEV) for k in ("pos_bert_input", "pos_mask", "pos_seg")}
    with torch.no_grad():
        got = r.test(d).cpu().numpy()
    e = rel_err(got, c["ref_scores"])
    scale = np.abs(got - c["ref_scores"]).max() / np.abs(c["ref_scores"]).max()
    print(name, dt, "max rel err", e.max(), "max abs err / max |score|", scale)
    if name == "base":   # the configuration the reference ships (13 hidden states, 1024-wide combine): the 1e-3 bar, element-wise
        assert e.max() <= (2e-2 if dt == "bf16" else 1e-3), (name, dt, e.max(), got, c["ref_scores"])
    else:
        # The mini fixtures put N(0, 0.3) weights on 128-192 [CLS] features of a 1-2 layer random encoder: the score is a sum of
        # terms of magnitude ~4 that cancel to ~1 or ~0.05, so a 5e-4 rounding of the 16-bit hidden states shows up as 1e-3..2e-2
        # of an individual small score.  They are path coverage (max / no [CLS] / single Linear / layer subsets): the error is
        # measured against the batch's score scale, and test_cedr_features_match_port_on_mini bounds the features themselves.
        assert scale <= (3e-2 if dt == "bf16" else 4e-3), (name, dt, scale, got, c["ref_scores"])


def test_cedr_features_match_port_on_mini():
    """The feature vector itself ([CLS] mean | 11 kernel features per selected hidden state), not only the score."""
    from oracle import bert_port
    from tests.helpers import load_cedr_case

    c, w, head, mus, sigmas = load_cedr_case("mini")
    r = _cedr_model(c, w, head, "fp16")
    d = [torch.from_numpy(c[k].astype(np.int64)) for k in ("pos_bert_input", "pos_mask", "pos_seg")]
    heads, layers = int(c["dims"][2]), int(c["dims"][1])
    n_in = head["combine.0.weight"].shape[1]
    eye = {"combine.0.weight": torch.eye(n_in), "combine.0.bias": torch.zeros(n_in)}       # the port with an identity head returns the features
    want = bert_port.cedr_knrm(w, eye, *d, heads, layers, int(c["maxqlen"]), [int(x) for x in c["simmat_layers"]], mus, sigmas, c["cls_mode"])
    want = want.view(d[0].shape[0], n_in).numpy()
    m = r.model
    with torch.no_grad():
        m(*[t.to(DEV) for t in d])
        mu, sigma = m.kernels.stacked()
        lin = m.combine[0]
        _, feats = m._engine.forward(*[t.to(DEV) for t in d], int(c["maxqlen"]), m._layers, mu, sigma, c["cls_mode"], lin.weight.detach().contiguous(),
                                     lin.bias.detach(), m.combine[1].weight.detach().view(-1), m.combine[1].bias.detach(), return_features=True)
    got = feats.cpu().numpy()
    H = int(c["dims"][0])
    assert np.abs(got[:, :H] - want[:, :H]).max() <= 2e-3 * np.abs(want[:, :H]).max()     # [CLS] rows of a 16-bit encoder
    assert np.abs(got[:, H:] - want[:, H:]).max() <= 2e-3 * np.abs(want[:, H:]).max(), np.abs(got[:, H:] - want[:, H:]).max()


@pytest.mark.parametrize("name", ["mini", "base"])
def test_cedr_length_buckets_give_the_same_features(name):
    """Passages regrouped by length (P = 1 calls, explicit first-passage query mask) vs one full-length call: same encoder
    arithmetic per passage, so the per-document features agree to the last bits of the fp32 sums."""
    from tests.helpers import load_cedr_case

    c, w, head, mus, sigmas = load_cedr_case(name)
    r = _cedr_model(c, w, head, "fp16")
    d = [torch.from_numpy(c[k].astype(np.int64)).to(DEV) for k in ("pos_bert_input", "pos_mask", "pos_seg")]
    m = r.model
    with torch.no_grad():
        m(*d)
        mu, sigma = m.kernels.stacked()
        lin = m.combine[0]
        w2 = m.combine[1].weight.detach().view(-1) if len(m.combine) == 2 else None
        b2 = m.combine[1].bias.detach() if len(m.combine) == 2 else None
        args = (int(c["maxqlen"]), m._layers, mu, sigma, c["cls_mode"], lin.weight.detach().contiguous(), lin.bias.detach(), w2, b2)
        s1, f1 = m._engine.forward(*d, *args, return_features=True, skip_padding=True)
        s0, f0 = m._engine.forward(*d, *args, return_features=True, skip_padding=False)
    assert torch.equal(f1[:, :int(c["dims"][0])], f0[:, :int(c["dims"][0])])                    # [CLS] rows: bit-identical
    assert (f1 - f0).abs().max().item() <= 1e-6 * f0.abs().max().item() + 1e-7, (f1 - f0).abs().max().item()
    assert (s1 - s0).abs().max().item() <= 1e-5 * s0.abs().max().item()


def test_cedr_knrm_on_an_electra_shaped_body():
    """ElectraModel = the same encoder without a pooler (CEDRKNRM.py:20-27 loads one by default): scores are those of the BERT body."""
    from capreolus_amd.reranker import CEDRKNRM
    from tests.helpers import CEDR_MUS, load_cedr_case

    c, w, head, mus, sigmas = load_cedr_case("mini")
    hidden, layers, heads, ffn, vocab, max_pos = (int(x) for x in c["dims"])
    P, S = c["pos_bert_input"].shape[1:]
    cfg = {"pretrained": dict(hidden=hidden, layers=layers, heads=heads, ffn=ffn, vocab=vocab, max_pos=max_pos, pooler=False), "mus": CEDR_MUS,
           "simmat_layers": [int(x) for x in c["simmat_layers"]], "combine_hidden": int(c["combine_hidden"]), "cls": c["cls_mode"]}
    r = CEDRKNRM(cfg, SimpleNamespace(config={"numpassages": P, "maxseqlen": S, "maxqlen": int(c["maxqlen"])}))
    m = r.build_model()
    sd = {k: v for k, v in w.items() if "pooler" not in k and not k.startswith("classifier")}
    sd.update(head)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    m.to(DEV).eval()
    d = {k: torch.from_numpy(c[k].astype(np.int64)).to(DEV) for k in ("pos_bert_input", "pos_mask", "pos_seg")}
    with torch.no_grad():
        got = r.test(d).cpu().numpy()
        want = _cedr_model(c, w, head, "fp16").test(d).cpu().numpy()
    assert np.array_equal(got, want)


# ---- the code paths the benchmark runs: persistent kernels with far more tiles / items than CUs -------------------------------
# BASELINE.json configs[3] encodes 256-passage micro-batches (M = 65,536 rows): 3,072 tiles per FFN1 launch and 3,072 attention
# items per layer on 256 CUs, i.e. the cross-tile fill chaining of gemm_pingpong_kernel and the next-item prefetch of
# attention_persistent_kernel, which the small shapes above never reach.

BIG_M = 65536


def _big_operands(M, N, K, seed, tdt):
    g = torch.Generator(device=DEV).manual_seed(seed)
    A = (torch.randn((M, K), generator=g, device=DEV) * 0.5).to(tdt)
    # every row tile and every column tile differ (a tile written to / read from the wrong place cannot cancel out)
    A += (torch.arange(M, device=DEV)[:, None] % 509).to(tdt) * 1e-3
    W = (torch.randn((N, K), generator=g, device=DEV) * 0.05 + torch.arange(N, device=DEV)[:, None] * 2e-4).to(tdt)
    return g, A, W


def _assert_one_ulp_apart(got, ref, tdt, what, frac=2e-2):
    """Two kernels that accumulate the same products in another order (k in steps of 32 on 16x16x32 MFMAs against steps of 16): fp32 sums
    that differ in their last bits, i.e. 16-bit results that are equal or - where a sum sat next to a rounding boundary - ONE unit in the
    last place apart, and that only for a small fraction of the elements."""
    a, b = got.view(torch.int16).int(), ref.view(torch.int16).int()
    # (adjacent finite values of one sign differ by 1 as integers; across zero both are tiny: compare the values there)
    d = (a - b).abs()
    same_sign = (a ^ b) >= 0
    bad = same_sign & (d > 1)
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements more than one ulp apart"
    cross = ~same_sign
    if bool(cross.any()):
        assert float((got.float() - ref.float())[cross].abs().max()) < 1e-3, f"{what}: values of opposite sign that are not both tiny"
    assert float((d != 0).float().mean()) < frac, f"{what}: {float((d != 0).float().mean()):.4f} of the elements differ"



def _assert_close_big(got, ref, rtol, atol, what):
    """row-blocked comparison: 65,536 x 3,072 fp32 temporaries stay below a few GB"""
    M = ref.shape[0]
    for lo in range(0, M, 16384):
        torch.testing.assert_close(got[lo:lo + 16384].float(), ref[lo:lo + 16384], rtol=rtol, atol=atol, msg=lambda m: f"{what} rows {lo}..: {m}")


@pytest.mark.parametrize("N,K,epi", [(768, 768, 0), (2304, 768, 0), (3072, 768, 1), (768, 3072, 0)])
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_gemm_at_benchmark_scale(N, K, epi, dt):
    M = BIG_M
    tdt, code, rtol = TDT[dt]
    g, A, W = _big_operands(M, N, K, N + K + epi, tdt)
    bias = torch.randn(N, generator=g, device=DEV)
    out = torch.empty((M, N), dtype=tdt, device=DEV)
    assert _lib.load().capamd_bert_gemm(_p(A), _p(W), _p(bias), M, N, K, epi, None, _p(out), code, _stream()) == 0
    ref = A.float() @ W.float().t() + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    _assert_close_big(out, ref, rtol, 2e-2 if dt == "bf16" else 3e-3, f"gemm {N}x{K} epi {epi}")
    # chunk-major in and out (the layout the encoder runs on): same numbers, other addresses
    out_cm, A_cm = torch.empty(M * N, dtype=tdt, device=DEV), _to_cm(A)
    assert _lib.load().capamd_bert_gemm(_p(A_cm), _p(W), _p(bias), M, N, K, epi | 0x300, None, _p(out_cm), code, _stream()) == 0
    assert torch.equal(_from_cm(out_cm, M, N), out)
    # weights chunk-major too -> the 4-wave ring kernel (bert_gemm_ring.h): same accumulation order, identical bits
    out_ring, W_cm = torch.empty(M * N, dtype=tdt, device=DEV), _to_cm(W)
    for rows256 in (0, 0x1800):         # the 128-row tile (two workgroups per CU) and the 256-row tile (one) on 32x32x16 MFMAs
        out_ring.zero_()
        assert _lib.load().capamd_bert_gemm(_p(A_cm), _p(W_cm), _p(bias), M, N, K, epi | 0x700 | rows256, None, _p(out_ring), code, _stream()) == 0
        assert torch.equal(out_ring, out_cm)
    # the 256-row tile on 16x16x32 MFMAs (bert_gemm_ring16.h, round 5): k accumulated in steps of 32 - the reference's numbers to the
    # same tolerance, the other kernels' to one unit in the last place
    out_ring.zero_()
    assert _lib.load().capamd_bert_gemm(_p(A_cm), _p(W_cm), _p(bias), M, N, K, epi | 0xF00, None, _p(out_ring), code, _stream()) == 0
    _assert_close_big(_from_cm(out_ring, M, N), ref, rtol, 2e-2 if dt == "bf16" else 3e-3, f"gemm {N}x{K} epi {epi} on 16x16x32")
    _assert_one_ulp_apart(out_ring, out_cm, tdt, f"gemm {N}x{K} epi {epi}: 16x16x32 vs 32x32x16")
    # ... and its 128-row form (two workgroups per CU): the same accumulation order, the same bits
    out_ring128 = torch.zeros_like(out_ring)
    assert _lib.load().capamd_bert_gemm(_p(A_cm), _p(W_cm), _p(bias), M, N, K, epi | 0x2700, None, _p(out_ring128), code, _stream()) == 0
    assert torch.equal(out_ring128, out_ring)
    out_ring_rm = torch.empty((M, N), dtype=tdt, device=DEV)      # ... and its row-major (LDS-staged) epilogue
    assert _lib.load().capamd_bert_gemm(_p(A_cm), _p(W_cm), _p(bias), M, N, K, epi | 0x600, None, _p(out_ring_rm), code, _stream()) == 0
    assert torch.equal(out_ring_rm, out)


@pytest.mark.parametrize("N,K,epi", [(2304, 768, 0), (3072, 768, 1)])
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_gemm_folded_layernorm_consumer_at_benchmark_scale(N, K, epi, dt):
    M = BIG_M
    tdt, code, rtol = TDT[dt]
    g = torch.Generator(device=DEV).manual_seed(N + K)
    P = (torch.randn((M, K), generator=g, device=DEV) * 1.7 + torch.randn((M, 1), generator=g, device=DEV) * 0.8).to(tdt)
    W0 = torch.randn((N, K), generator=g, device=DEV) * 0.04 + torch.arange(N, device=DEV)[:, None] * 2e-4
    b = torch.randn(N, generator=g, device=DEV) * 0.3
    gamma = 1.0 + 0.2 * torch.randn(K, generator=g, device=DEV)
    beta = 0.1 * torch.randn(K, generator=g, device=DEV)
    Pf = P.float()
    mu = Pf.mean(1)
    rstd = torch.rsqrt(Pf.var(1, unbiased=False) + 1e-12)
    Wg = W0 * gamma[None, :]
    Ws = (Wg - Wg.mean(1, keepdim=True)).to(tdt)
    cs = Ws.float().sum(1)
    c = b + W0 @ beta
    mr = torch.stack([mu, rstd], 1).contiguous()
    out, P_cm = torch.empty(M * N, dtype=tdt, device=DEV), _to_cm(P)
    rc = _lib.load().capamd_bert_gemm_ln(_p(P_cm), _p(Ws), _p(c), M, N, K, epi | 0x300, _p(mu), _p(rstd), _p(mr), _p(cs), None, None, None, None,
                                         _p(out), code, _stream())
    assert rc == 0
    ref = ((Pf - mu[:, None]) * rstd[:, None] * gamma + beta) @ W0.t() + b
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    _assert_close_big(_from_cm(out, M, N), ref, 4 * rtol, 5e-2 if dt == "bf16" else 8e-3, f"folded-LN consumer {N}x{K}")
    out_ring, W_cm = torch.empty(M * N, dtype=tdt, device=DEV), _to_cm(Ws)      # the ring kernel: identical bits
    for rows256 in (0, 0x1800):
        out_ring.zero_()
        rc = _lib.load().capamd_bert_gemm_ln(_p(P_cm), _p(W_cm), _p(c), M, N, K, epi | 0x700 | rows256, _p(mu), _p(rstd), _p(mr), _p(cs), None, None, None,
                                             None, _p(out_ring), code, _stream())
        assert rc == 0 and torch.equal(out_ring, out)
    out_ring.zero_()          # the 256-row tile on 16x16x32 MFMAs
    rc = _lib.load().capamd_bert_gemm_ln(_p(P_cm), _p(W_cm), _p(c), M, N, K, epi | 0xF00, _p(mu), _p(rstd), _p(mr), _p(cs), None, None, None, None,
                                         _p(out_ring), code, _stream())
    assert rc == 0
    _assert_close_big(_from_cm(out_ring, M, N), ref, 4 * rtol, 5e-2 if dt == "bf16" else 8e-3, f"folded-LN consumer {N}x{K} on 16x16x32")
    _assert_one_ulp_apart(out_ring, out, tdt, f"folded-LN consumer {N}x{K}: 16x16x32 vs 32x32x16")
    out_ring128 = torch.zeros_like(out_ring)          # the 128-row form of the 16x16x32 kernel: identical bits
    rc = _lib.load().capamd_bert_gemm_ln(_p(P_cm), _p(W_cm), _p(c), M, N, K, epi | 0x2700, _p(mu), _p(rstd), _p(mr), _p(cs), None, None, None, None,
                                         _p(out_ring128), code, _stream())
    assert rc == 0 and torch.equal(out_ring128, out_ring)


@pytest.mark.parametrize("N,K", [(768, 768), (768, 3072)])
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_gemm_residual_stats_producer_at_benchmark_scale(N, K, dt):
    M = BIG_M
    tdt, code, rtol = TDT[dt]
    g, A, W = _big_operands(M, N, K, N + K + 5, tdt)
    bp = torch.randn(N, generator=g, device=DEV)
    R = (torch.randn((M, N), generator=g, device=DEV) * 2.0 + 0.5).to(tdt)
    gamma = 1.0 + 0.2 * torch.randn(N, generator=g, device=DEV)
    Rf = R.float()
    mu = Rf.mean(1)
    rstd = torch.rsqrt(Rf.var(1, unbiased=False) + 1e-12)
    mr = torch.stack([mu, rstd], 1).contiguous()
    out = torch.empty(M * N, dtype=tdt, device=DEV)
    part = torch.zeros((M, N // 64, 2), device=DEV)
    A_cm, R_cm = _to_cm(A), _to_cm(R)       # (named: a temporary's memory could be handed to the next temporary before the launch)
    rc = _lib.load().capamd_bert_gemm_ln(_p(A_cm), _p(W), _p(bp), M, N, K, 5 | 0x300, None, None, None, None, _p(R_cm), _p(mr), _p(gamma), _p(part),
                                         _p(out), code, _stream())
    assert rc == 0
    ref = A.float() @ W.float().t() + bp + (Rf - mu[:, None]) * rstd[:, None] * gamma
    got = _from_cm(out, M, N).float()
    _assert_close_big(got, ref, rtol, 2e-2 if dt == "bf16" else 3e-3, f"residual+stats producer {N}x{K}")
    want = torch.stack([got.reshape(M, N // 64, 64).sum(2), (got * got).reshape(M, N // 64, 64).sum(2)], 2)
    torch.testing.assert_close(part, want, rtol=1e-4, atol=1e-3)
    out_ring, part_ring, W_cm = torch.empty(M * N, dtype=tdt, device=DEV), torch.zeros((M, N // 64, 2), device=DEV), _to_cm(W)
    for rows256 in (0, 0x1800):
        out_ring.zero_(), part_ring.zero_()
        rc = _lib.load().capamd_bert_gemm_ln(_p(A_cm), _p(W_cm), _p(bp), M, N, K, 5 | 0x700 | rows256, None, None, None, None, _p(R_cm), _p(mr), _p(gamma),
                                             _p(part_ring), _p(out_ring), code, _stream())
        assert rc == 0 and torch.equal(out_ring, out) and torch.equal(part_ring, part)      # the ring kernel: identical bits
    out_ring.zero_(), part_ring.zero_()          # the 256-row tile on 16x16x32 MFMAs: its own sums, statistics of what IT stored
    rc = _lib.load().capamd_bert_gemm_ln(_p(A_cm), _p(W_cm), _p(bp), M, N, K, 5 | 0xF00, None, None, None, None, _p(R_cm), _p(mr), _p(gamma),
                                         _p(part_ring), _p(out_ring), code, _stream())
    assert rc == 0
    got16 = _from_cm(out_ring, M, N).float()
    _assert_close_big(got16, ref, rtol, 2e-2 if dt == "bf16" else 3e-3, f"residual+stats producer {N}x{K} on 16x16x32")
    _assert_one_ulp_apart(out_ring, out, tdt, f"residual+stats producer {N}x{K}: 16x16x32 vs 32x32x16")
    want16 = torch.stack([got16.reshape(M, N // 64, 64).sum(2), (got16 * got16).reshape(M, N // 64, 64).sum(2)], 2)
    torch.testing.assert_close(part_ring, want16, rtol=1e-4, atol=1e-3)
    out_ring128, part_ring128 = torch.zeros_like(out_ring), torch.zeros_like(part_ring)      # the 128-row form: identical bits
    rc = _lib.load().capamd_bert_gemm_ln(_p(A_cm), _p(W_cm), _p(bp), M, N, K, 5 | 0x2700, None, None, None, None, _p(R_cm), _p(mr), _p(gamma),
                                         _p(part_ring128), _p(out_ring128), code, _stream())
    assert rc == 0 and torch.equal(out_ring128, out_ring) and torch.equal(part_ring128, part_ring)


@pytest.mark.parametrize("M,N,K,epi", [(256, 256, 256, 0), (512, 768, 512, 1), (768, 256, 1024, 0), (256, 256, 0x800 | 256, 0), (768, 512, 0x800 | 320, 1), (768, 512, 320, 1)])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_ring_gemm_small_shapes(M, N, K, epi, dt):
    """The 4-wave ring kernel on a handful of tiles (fewer tiles than CUs, the minimum of 16 k-slices, a tile count that is not a
    multiple of 8) against fp32 torch."""
    tdt, code, rtol = TDT[dt]
    rows256, K = K & 0x800, K & 0x7ff           # (the 256-row tile variant is encoded in the K parameter of the test id)
    g = torch.Generator(device=DEV).manual_seed(M + N + K + epi)
    A = (torch.randn((M, K), generator=g, device=DEV) * 0.5).to(tdt)
    W = (torch.randn((N, K), generator=g, device=DEV) * 0.05 + torch.arange(N, device=DEV)[:, None] * 1e-3).to(tdt)
    bias = torch.randn(N, generator=g, device=DEV)
    A_cm, W_cm, out = _to_cm(A), _to_cm(W), torch.empty(M * N, dtype=tdt, device=DEV)
    assert _lib.load().capamd_bert_gemm(_p(A_cm), _p(W_cm), _p(bias), M, N, K, epi | 0x700 | rows256, None, _p(out), code, _stream()) == 0
    ref = A.float() @ W.float().t() + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    torch.testing.assert_close(_from_cm(out, M, N).float(), ref, rtol=rtol, atol=2e-2 if dt == "bf16" else 3e-3)
    # shapes / layouts the ring kernel does not take are refused, not mis-computed
    assert _lib.load().capamd_bert_gemm(_p(A_cm), _p(W_cm), _p(bias), M, N, 128, epi | 0x700, None, _p(out), code, _stream()) != 0
    assert _lib.load().capamd_bert_gemm(_p(A), _p(W_cm), _p(bias), M, N, K, epi | 0x400, None, _p(out), code, _stream()) != 0


@pytest.mark.parametrize("hidden,heads,npsg", [(768, 12, 400), (128, 2, 1024)])
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_qkv_attention_many_items(hidden, heads, npsg, dt):
    """S = 256 with 4,800 / 2,048 (passage, head) items on 256 CUs: every persistent workgroup walks several items, K / V^T of the next
    item arriving in the second LDS buffer while the current one is computed."""
    S = 256
    tdt, code, rtol = TDT[dt]
    atol = 2e-2 if dt == "bf16" else 3e-3
    g = torch.Generator(device=DEV).manual_seed(hidden + npsg)
    M = npsg * S
    x = torch.randn((M, hidden), generator=g, device=DEV).to(tdt)
    w = (torch.randn((3 * hidden, hidden), generator=g, device=DEV) * 0.06).to(tdt)
    b = torch.randn(3 * hidden, generator=g, device=DEV) * 0.1
    lens = torch.randint(5, S + 1, (npsg,), generator=g, device=DEV)
    mask = (torch.arange(S, device=DEV)[None, :] < lens[:, None]).long()
    mask[::7, 9] = 0
    q, k, ctx = (torch.empty((M, hidden), dtype=tdt, device=DEV) for _ in range(3))
    vt = torch.empty((npsg * heads, 64, S), dtype=tdt, device=DEV)
    rc = _lib.load().capamd_bert_qkv_attention(_p(x), _p(w), _p(b), _p(mask), npsg, S, hidden, heads, _p(q), _p(k), _p(vt), _p(ctx), code, _stream())
    assert rc == 0
    for lo in range(0, npsg, 100):      # reference in blocks of 100 passages
        hi = min(lo + 100, npsg)
        n = hi - lo
        qkv = x[lo * S:hi * S].float() @ w.float().t() + b
        qr, kr, vr = (t.view(n, S, heads, 64).transpose(1, 2) for t in qkv.split(hidden, dim=1))
        torch.testing.assert_close(q[lo * S:hi * S].float().view(n, S, heads, 64).transpose(1, 2), qr / 8, rtol=rtol, atol=atol)
        torch.testing.assert_close(k[lo * S:hi * S].float().view(n, S, heads, 64).transpose(1, 2), kr, rtol=rtol, atol=atol)
        torch.testing.assert_close(vt[lo * heads:hi * heads].float().view(n, heads, 64, S).transpose(2, 3), vr, rtol=rtol, atol=atol)
        qb = q[lo * S:hi * S].float().view(n, S, heads, 64).transpose(1, 2)
        kb = k[lo * S:hi * S].float().view(n, S, heads, 64).transpose(1, 2)
        vb = vt[lo * heads:hi * heads].float().view(n, heads, 64, S).transpose(2, 3)
        att = torch.softmax(qb @ kb.transpose(-1, -2) + (1.0 - mask[lo:hi].float()).view(n, 1, 1, S) * torch.finfo(torch.float32).min, dim=-1)
        ref = (att @ vb).transpose(1, 2).reshape(n * S, hidden)
        torch.testing.assert_close(ctx[lo * S:hi * S].float(), ref, rtol=2e-2 if dt == "bf16" else 3e-3, atol=atol)


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_bert_base_benchmark_microbatch_matches_fixture_sized_runs(dt):
    """A 256-passage micro-batch (the size bench.py times: 65,536 rows, every GEMM > 256 tiles) gives BIT-IDENTICAL passage logits
    to the same passages encoded 12 at a time - the size at which tests/golden/bert_base.npz pins the engine to the reference.
    The first 12 passages ARE that fixture's passages and weights, so the chain reference -> 12-passage run -> benchmark-sized run
    is closed: their logits also stay within the fixture's tolerance of the reference."""
    from capreolus_amd import synthetic

    c = load_bert_case("base")
    B0, P, S = c["pos_bert_input"].shape                      # 3 documents x 4 passages x 256 tokens
    extra = synthetic.make_bert_passages(np.random.RandomState(77), 61, P, S, vocab=c["vocab"])
    d = {k: torch.cat([c[k], torch.as_tensor(extra[k])]).to(DEV) for k in ("pos_bert_input", "pos_mask", "pos_seg")}   # 64 docs = 256 passages
    r = _model(c, "max", dt)
    eng_args = (d["pos_bert_input"], d["pos_mask"], d["pos_seg"], "max")
    with torch.no_grad():
        r.test({k: v[:B0] for k, v in d.items()})
        eng = r.model._engine
        eng.n_streams = 1
        eng.microbatch = 256
        s256, pl256 = eng.forward(*eng_args, return_passage_logits=True, skip_padding=False)
        eng.microbatch = 12
        s12, pl12 = eng.forward(*eng_args, return_passage_logits=True, skip_padding=False)
    assert torch.equal(pl256, pl12) and torch.equal(s256, s12)
    tol = BF16_E2E_TOL if dt == "bf16" else FP16_E2E_TOL
    assert rel_err(pl256[:B0 * P].cpu().numpy(), c["ref_passage_logits"][:, 1]).max() <= tol
    # and 1,000 passages (four micro-batches, the last one ragged) through the multi-stream path the engine uses by default (2 slices) and
    # the bench's (3): slices on their own streams and workspaces
    big = {k: torch.cat([v] * 4)[:250] for k, v in d.items()}
    for ns in (2, 3):
        with torch.no_grad():
            eng.microbatch, eng.n_streams = 256, ns
            sb, plb = eng.forward(big["pos_bert_input"], big["pos_mask"], big["pos_seg"], "max", return_passage_logits=True, skip_padding=False)
        assert torch.equal(plb[:256], pl256) and torch.equal(plb[256:512], pl256) and torch.equal(plb[768:], pl256[:232])


@pytest.mark.parametrize("S,n_passages", [(32, 1), (32, 13), (96, 5), (160, 7), (224, 3), (64, 5), (128, 3)])
@pytest.mark.parametrize("dt", ["fp16"])
def test_ragged_passage_counts(S, n_passages, dt):
    """Passage counts whose rows do not fill whole 64- / 256-row GEMM tiles (an odd count at S = 32, 96, 160, 224; a single
    32-token passage): the library pads the micro-batch itself.  Every passage must score exactly as it does inside a batch that
    needs no padding, and the scores must match the fp32 port."""
    from capreolus_amd.engine import BertEngine
    from oracle import bert_port

    heads, layers, vocab = 4, 2, 500
    w = bert_port.random_weights(hidden=256, layers=layers, heads=heads, ffn=512, vocab=vocab, max_pos=S, seed=S + n_passages)
    g = torch.Generator().manual_seed(S * 3 + n_passages)
    n_full = 16                                               # 16 passages of any supported S are whole 256-row tiles
    ids = torch.randint(1, vocab, (n_full, 1, S), generator=g)
    lens = torch.randint(8, S + 1, (n_full, 1), generator=g)
    mask = (torch.arange(S)[None, None, :] < lens[:, :, None]).long()
    seg = ((torch.arange(S)[None, None, :] >= 6) & (mask > 0)).long()
    ids = ids * mask
    eng = BertEngine({k: v.to(DEV) for k, v in w.items()}, heads, compute_dtype=dt, skip_padding=False)
    with torch.no_grad():
        full = eng.forward(ids.to(DEV), mask.to(DEV), seg.to(DEV), "max")
        n = n_passages
        part = eng.forward(ids[:n].to(DEV), mask[:n].to(DEV), seg[:n].to(DEV), "max")
        eng.microbatch = 1                                    # ... and with a ragged LAST micro-batch (S = 32: 8 passages per micro-batch)
        split = eng.forward(ids[:n].to(DEV), mask[:n].to(DEV), seg[:n].to(DEV), "max")
    assert torch.equal(part, full[:n]) and torch.equal(split, full[:n])
    ref = bert_port.maxp(w, ids[:n], mask[:n], seg[:n], heads, layers, "max")
    # (a 2-layer random model has small scores: measured against the batch's score scale, as for the mini CEDR fixtures)
    assert np.abs(part.cpu().numpy() - ref.numpy()).max() <= 4e-3 * max(1.0, float(ref.abs().max()))
