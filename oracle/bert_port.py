"""CPU restatement of the BERT-MaxP scoring path in fp32 (TEST INFRASTRUCTURE ONLY — see
oracle/__init__.py).

What it restates
  * PTBERTMaxP_Class.predict_step (reference capreolus/reranker/ptBERTMaxP.py:67-96): passage mask
    statistics, flatten to [B*P, S], score every passage, take logit 1, pool (max | first | sum | avg).
  * the encoder it calls at ptBERTMaxP.py:82, `transformers.BertForSequenceClassification`
    (third-party; the reference pins transformers~=4.9.2 in setup.py:73, this container has 5.15.0 —
    same published BERT arithmetic): embeddings = word + position + token_type -> LayerNorm(eps 1e-12);
    12 x [Q/K/V projections, softmax(QK^T/sqrt(64) + additive pad mask) V, output projection,
    +residual, LayerNorm, 768->3072 erf-GELU ->768, +residual, LayerNorm]; pooler tanh(W h_CLS + b);
    classifier 768 -> 2.
Pinned by tests/test_oracle_golden.py against golden vectors produced by the reference module
driving the HF model (tests/golden/make_golden_bert.py).

Weights are a plain dict with the HF state_dict key names, values torch fp32.
"""
import math

import torch
import torch.nn.functional as F


def random_weights(hidden=768, layers=12, heads=12, ffn=3072, vocab=30522, max_pos=512, type_vocab=2, seed=0, std=0.05):
    """Seeded stand-in for a checkpoint (there is no network for real ones).  Wider than HF's 0.02
    init and with non-trivial LayerNorm/bias terms so that every term of the forward matters."""
    g = torch.Generator().manual_seed(seed)

    def n(*shape, s=std):
        return torch.randn(*shape, generator=g) * s

    w = {
        "bert.embeddings.word_embeddings.weight": n(vocab, hidden),
        "bert.embeddings.position_embeddings.weight": n(max_pos, hidden),
        "bert.embeddings.token_type_embeddings.weight": n(type_vocab, hidden),
        "bert.embeddings.LayerNorm.weight": 1.0 + n(hidden, s=0.1),
        "bert.embeddings.LayerNorm.bias": n(hidden, s=0.1),
        "bert.pooler.dense.weight": n(hidden, hidden),
        "bert.pooler.dense.bias": n(hidden, s=0.1),
        "classifier.weight": n(2, hidden, s=0.2),
        "classifier.bias": n(2, s=0.1),
    }
    w["bert.embeddings.word_embeddings.weight"][0] = 0  # padding_idx row
    for i in range(layers):
        p = f"bert.encoder.layer.{i}."
        for name in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            w[p + name + ".weight"] = n(hidden, hidden)
            w[p + name + ".bias"] = n(hidden, s=0.1)
        w[p + "intermediate.dense.weight"] = n(ffn, hidden)
        w[p + "intermediate.dense.bias"] = n(ffn, s=0.1)
        w[p + "output.dense.weight"] = n(hidden, ffn)
        w[p + "output.dense.bias"] = n(hidden, s=0.1)
        for ln in ("attention.output.LayerNorm", "output.LayerNorm"):
            w[p + ln + ".weight"] = 1.0 + n(hidden, s=0.1)
            w[p + ln + ".bias"] = n(hidden, s=0.1)
    return w


def _ln(x, w, b, eps=1e-12):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def encode_passages(w, input_ids, attention_mask, token_type_ids, heads, layers):
    """[N, S] int64 x3 -> logits [N, 2] (BertForSequenceClassification.forward in eval mode)."""
    N, S = input_ids.shape
    H = w["bert.embeddings.word_embeddings.weight"].shape[1]
    dh = H // heads
    x = (w["bert.embeddings.word_embeddings.weight"][input_ids]
         + w["bert.embeddings.position_embeddings.weight"][:S].unsqueeze(0)
         + w["bert.embeddings.token_type_embeddings.weight"][token_type_ids])
    x = _ln(x, w["bert.embeddings.LayerNorm.weight"], w["bert.embeddings.LayerNorm.bias"])
    bias = (1.0 - attention_mask.float()).view(N, 1, 1, S) * torch.finfo(torch.float32).min
    for i in range(layers):
        p = f"bert.encoder.layer.{i}."

        def lin(name, t):
            return t @ w[p + name + ".weight"].t() + w[p + name + ".bias"]

        q = lin("attention.self.query", x).view(N, S, heads, dh).transpose(1, 2)
        k = lin("attention.self.key", x).view(N, S, heads, dh).transpose(1, 2)
        v = lin("attention.self.value", x).view(N, S, heads, dh).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + bias, dim=-1)
        ctx = (a @ v).transpose(1, 2).reshape(N, S, H)
        x = _ln(lin("attention.output.dense", ctx) + x, w[p + "attention.output.LayerNorm.weight"],
                w[p + "attention.output.LayerNorm.bias"])
        h = F.gelu(lin("intermediate.dense", x))  # erf GELU
        x = _ln(lin("output.dense", h) + x, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"])
    pooled = torch.tanh(x[:, 0] @ w["bert.pooler.dense.weight"].t() + w["bert.pooler.dense.bias"])
    return pooled @ w["classifier.weight"].t() + w["classifier.bias"]


def maxp(w, doc_input, doc_mask, doc_seg, heads, layers, aggregation="max", chunk=64):
    """PTBERTMaxP_Class.predict_step (ptBERTMaxP.py:67-96): [B, P, S] int64 x3 -> [B] fp32."""
    B, P, S = doc_input.shape
    passage_position = (doc_mask * doc_seg).sum(dim=-1)           # :75
    passage_mask = (passage_position > 5).long()                  # :76
    flat = [t.reshape(B * P, S) for t in (doc_input, doc_mask, doc_seg)]
    outs = []
    with torch.no_grad():
        for lo in range(0, B * P, chunk):
            outs.append(encode_passages(w, flat[0][lo:lo + chunk], flat[1][lo:lo + chunk], flat[2][lo:lo + chunk], heads, layers)[:, 1])
    s = torch.cat(outs).reshape(B, P)                             # :82-83
    if aggregation == "max":
        return s.max(dim=1)[0]
    if aggregation == "first":
        return s[:, 0]
    if aggregation == "sum":
        return torch.sum(passage_mask * s, dim=1)
    if aggregation == "avg":
        return torch.sum(passage_mask * s, dim=1) / torch.sum(passage_mask)  # batch-wide denominator (:92)
    raise ValueError("Unknown aggregation method: {}".format(aggregation))
