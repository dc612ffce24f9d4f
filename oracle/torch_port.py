"""Second CPU restatement: the reference's *op sequence* on PyTorch CPU tensors.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Where interaction_oracle.c fixes an
arithmetic order to agree bit-for-bit with the GPU, this file keeps the shape of what the
reference executes on the host — embedding gather, bmm, norms, elementwise kernels — so that
bench.py's cpu_baseline can time "the reference CPU path"  with ATen's own
threading.  Pinned against the golden vectors in tests/test_oracle_golden.py.
"""
import torch


def similarity(emb, q, d):
    """SimilarityMatrix.forward (reranker/common.py:143-182) -> [B, Q, L]."""
    B, Q = q.shape
    L = d.shape[1]
    qo, do = q.clamp(max=0).view(B, Q, 1), d.clamp(max=0).view(B, 1, L)   # OOV part (:179)
    exact = ((qo == do) & (qo != 0) & (do != 0)).float()                   # exact_match_matrix (:155-158)
    qi, di = q.clamp(min=0), d.clamp(min=0)                                # in-vocab part (:180)
    a, b = emb[qi], emb[di]                                                # embedding gather (:161)
    den = (a.norm(p=2, dim=2) + 1e-9).view(B, Q, 1) * (b.norm(p=2, dim=2) + 1e-9).view(B, 1, L)
    cos = a.bmm(b.transpose(1, 2)) / den                                   # (:162-166)
    cos = cos.masked_fill((qi == 0).view(B, Q, 1) | (di == 0).view(B, 1, L), 0.0)  # remove_padding (:149-153)
    return exact + cos


def knrm(emb, q, d, mu, sigma, w1, b1, w2=None, b2=None, scoretanh=False):
    """KNRM_class.forward (reranker/KNRM.py:39-55) -> [B]."""
    sim = similarity(emb, q, d)
    adj = sim.unsqueeze(1) - mu.view(1, -1, 1, 1)
    kern = torch.exp(-0.5 * adj * adj / sigma.view(1, -1, 1, 1) / sigma.view(1, -1, 1, 1))  # common.py:232-234
    pooled = kern.sum(dim=3)                                               # KNRM.py:50
    mask = (sim.sum(dim=2) != 0.0).unsqueeze(1)                            # KNRM.py:51
    feats = torch.where(mask, (pooled + 1e-6).log(), torch.zeros_like(pooled)).sum(dim=2)  # :52-53
    if w2 is None:
        s = feats @ w1.t() + b1
    else:
        s = torch.tanh(feats @ w1.t() + b1) @ w2.t() + b2
    if scoretanh:
        s = torch.tanh(s)
    return s.view(-1)


def drmm(emb, q, d, idf, nbins, hist_type, gate_type, gate_w, w1, b1, w2, b2, out_w, out_b):
    """DRMM_class.forward (reranker/DRMM.py:101-116) -> [B]."""
    sim = similarity(emb, q, d)
    sim = sim + (d == 0).float().unsqueeze(1) * 1e7                        # DRMM.py:57
    edges = torch.linspace(-1, 1, nbins + 1)[1:]
    cum = (sim.unsqueeze(-1) < edges).sum(dim=2).float()                   # DRMM.py:62-65
    hist = torch.cat([cum[..., :1], cum[..., 1:] - cum[..., :-1],          # :68-69
                      ((sim > 0.999) & (sim < 1.001)).sum(dim=2, keepdim=True).float()], dim=-1) + 1  # :66, :71
    if hist_type == "NH":
        hist = hist / hist.sum(dim=-1, keepdim=True)
    elif hist_type == "LCH":
        hist = hist.log()
    z = torch.tanh(torch.tanh(hist @ w1.t() + b1) @ w2.view(-1, 1) + b2).squeeze(-1)   # ffw (:25)
    if gate_type == "IDF":
        gl = idf * gate_w.view(-1)[0]
    else:
        gl = emb[q.clamp(min=0)] @ gate_w.view(-1)
    gl = gl + (q == 0).float() * -1e7                                      # :90
    g = torch.softmax(gl, dim=1)
    return ((g * z).sum(dim=1) * out_w.view(-1)[0] + out_b.view(-1)[0]).view(-1)
