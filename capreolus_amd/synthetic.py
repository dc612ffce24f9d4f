"""Synthetic candidate lists in the layout the reference extractors emit.

Shapes / dtypes follow EmbedText.id2vec (reference capreolus/extractor/embedtext.py:128-162):
``query`` int64 [B, maxqlen], ``posdoc`` int64 [B, maxdoclen], ``query_idf`` float32 [B, maxqlen];
pad id 0, OOV terms carry *negative* ids (embedtext.py:118-123), pads trail the real terms
(capreolus/utils/common.py:99-111).  BertPassage layout follows bertpassage.py:268-346.

Distributions are the ones SURVEY.md §8(d) fixes for BASELINE.json configs 2-5.  Two
generators: a numpy one (bit-stable across machines; used for golden fixtures and parity tests)
and a torch one that builds the big benchmark batches directly in device memory.
"""
import numpy as np

__all__ = [
    "make_embeddings",
    "zipf_ids",
    "make_candidate_list",
    "make_candidate_list_torch",
    "make_bert_passages",
]


def make_embeddings(vocab, dim, seed=0, scale=0.4):
    """fp32 [vocab, dim] table, row 0 = zeros (pad) as extractor/common.py:38-40 builds it."""
    rs = np.random.RandomState(seed)
    emb = (rs.standard_normal((vocab, dim)) * scale).astype(np.float32)
    emb[0] = 0.0
    return emb


def _zipf_cdf(vocab, a):
    w = np.arange(1, vocab, dtype=np.float64) ** (-a)
    c = np.cumsum(w)
    return c / c[-1]


def zipf_ids(rs, n, vocab, a=1.1):
    """n ids in [1, vocab) with P(k) ~ k^-a (inverse-CDF sampling; exact truncation)."""
    cdf = _zipf_cdf(vocab, a)
    u = rs.random_sample(n)
    return (np.searchsorted(cdf, u, side="left") + 1).astype(np.int64).clip(1, vocab - 1)


def make_candidate_list(
    rs, n_pairs, vocab, maxqlen=4, maxdoclen=800, same_query=True, oov_frac=0.02, oov_range=5000,
    match_frac=0.30, query_oov_frac=0.0, zipf_a=1.1, idf=True,
):
    """One candidate list (n_pairs docs for one query when same_query) as numpy arrays.

    Returns dict(query [B,Q] i64, posdoc [B,L] i64, query_idf [B,Q] f32).
    """
    B, Q, L = n_pairs, maxqlen, maxdoclen
    nq = 1 if same_query else B
    qlen = rs.randint(1, Q + 1, size=nq)
    q = np.zeros((nq, Q), dtype=np.int64)
    for i in range(nq):
        q[i, : qlen[i]] = zipf_ids(rs, qlen[i], vocab, zipf_a)
        if query_oov_frac > 0:
            m = rs.random_sample(qlen[i]) < query_oov_frac
            q[i, : qlen[i]][m] = -rs.randint(1, oov_range + 1, size=int(m.sum()))
    if same_query:
        q = np.repeat(q, B, axis=0)
    lo = min(20, L)
    dlen = np.clip(np.exp(rs.normal(5.5, 0.8, size=B)), lo, L).astype(np.int64)
    d = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        n = int(dlen[b])
        toks = zipf_ids(rs, n, vocab, zipf_a)
        m = rs.random_sample(n) < oov_frac
        toks[m] = -rs.randint(1, oov_range + 1, size=int(m.sum()))
        if rs.random_sample() < match_frac:
            nm = rs.randint(1, 6)
            pos = rs.randint(0, n, size=nm)
            real_q = q[b][q[b] != 0]
            toks[pos] = real_q[rs.randint(0, len(real_q), size=nm)]
        d[b, :n] = toks
    qidf = np.zeros((B, Q), dtype=np.float32)
    if idf:
        qidf = np.where(q != 0, rs.uniform(0.5, 8.0, size=(B, Q)), 0.0).astype(np.float32)
    return {"query": q, "posdoc": d, "query_idf": qidf}


def make_candidate_list_torch(
    n_queries, docs_per_query, vocab, device, seed=1, maxqlen=4, maxdoclen=800, oov_frac=0.02,
    oov_range=5000, match_frac=0.30, zipf_a=1.1, idf=True, uniform_ids=False,
):
    """Benchmark-size candidate lists built on `device` (same distributions as the numpy generator).

    Returns dict of tensors query [N,Q] i64, posdoc [N,L] i64, query_idf [N,Q] f32 with
    N = n_queries*docs_per_query, laid out query-major (a query's candidates are contiguous).
    """
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    N, Q, L = n_queries * docs_per_query, maxqlen, maxdoclen

    if uniform_ids:
        def draw(shape):
            return torch.randint(1, vocab, shape, generator=g, device=device, dtype=torch.int64)
    else:
        w = torch.arange(1, vocab, device=device, dtype=torch.float64) ** (-zipf_a)
        cdf = torch.cumsum(w, 0)
        cdf = (cdf / cdf[-1]).float()

        def draw(shape):
            u = torch.rand(shape, generator=g, device=device)
            return (torch.searchsorted(cdf, u) + 1).clamp_(1, vocab - 1)

    qlen = torch.randint(1, Q + 1, (n_queries, 1), generator=g, device=device)
    q = draw((n_queries, Q))
    q = torch.where(torch.arange(Q, device=device)[None, :] < qlen, q, torch.zeros_like(q))
    q = q.repeat_interleave(docs_per_query, dim=0)

    lo = min(20, L)
    dlen = torch.exp(torch.randn((N, 1), generator=g, device=device) * 0.8 + 5.5).clamp_(lo, L).long()
    d = draw((N, L))
    oov = torch.rand((N, L), generator=g, device=device) < oov_frac
    d = torch.where(oov, -torch.randint(1, oov_range + 1, (N, L), generator=g, device=device), d)
    # inject query terms into ~match_frac of the documents (up to 5 positions)
    has = torch.rand((N, 1), generator=g, device=device) < match_frac
    pos = (torch.rand((N, 5), generator=g, device=device) * dlen).long()
    which = torch.randint(0, Q, (N, 5), generator=g, device=device)
    qterm = torch.gather(q, 1, which)
    use = has & (qterm != 0) & (torch.rand((N, 5), generator=g, device=device) < 0.6)
    cur = torch.gather(d, 1, pos)
    d.scatter_(1, pos, torch.where(use, qterm, cur))
    d = torch.where(torch.arange(L, device=device)[None, :] < dlen, d, torch.zeros_like(d))
    if idf:
        qidf = torch.rand((N, Q), generator=g, device=device) * 7.5 + 0.5
        # (an idf vector is a property of the QUERY, embedtext.py:131-135: every candidate of a query carries the query's - its first draw)
        qidf = qidf.view(n_queries, docs_per_query, Q)[:, :1].expand(n_queries, docs_per_query, Q).reshape(N, Q)
        qidf = torch.where(q != 0, qidf, torch.zeros_like(qidf))
    else:
        qidf = torch.zeros((N, Q), device=device)
    return {"query": q.contiguous(), "posdoc": d.contiguous(), "query_idf": qidf.float().contiguous()}


def make_bert_passages(rs, n_docs, numpassages=4, maxseqlen=256, vocab=30522, empty_frac=0.15, same_query=True, body_range=(40, 240)):
    """BertPassage-shaped inputs (bertpassage.py:268-284, 313-325): int64 [B, P, S] x3.

    `[CLS] q [SEP] psg [SEP] [PAD]...`; mask = 1 on non-pad; seg = 0 for len(q)+2 tokens then 1
    *to the end including pads* (reference test_extractor.py:708-713).  Empty passages are
    `[CLS] q [SEP] [PAD] [SEP]` (bertpassage.py:229, 165).
    """
    CLS, SEP, PAD = 101, 102, 0
    B, P, S = n_docs, numpassages, maxseqlen
    inp = np.zeros((B, P, S), dtype=np.int64)
    mask = np.zeros((B, P, S), dtype=np.int64)
    seg = np.zeros((B, P, S), dtype=np.int64)
    qmax = max(1, min(12, S // 4))
    tlo = min(1000, vocab // 4)  # BERT's first ~1000 ids are special/unused tokens
    q_shared = rs.randint(tlo, vocab, size=rs.randint(min(3, qmax), qmax + 1))
    for b in range(B):
        q = q_shared if same_query else rs.randint(tlo, vocab, size=rs.randint(min(3, qmax), qmax + 1))
        nq = len(q)
        for p in range(P):
            head = [CLS] + list(q) + [SEP]
            room = S - len(head) - 1
            if rs.random_sample() < empty_frac and p > 0:
                body = [PAD]
            else:
                plo, phi = min(body_range[0], room), min(body_range[1], room)
                body = list(rs.randint(tlo, vocab, size=rs.randint(plo, phi + 1)))
            toks = head + body + [SEP]
            n = len(toks)
            inp[b, p, :n] = toks
            mask[b, p, :n] = [0 if t == PAD else 1 for t in toks]
            seg[b, p, nq + 2:] = 1
    return {"pos_bert_input": inp, "pos_mask": mask, "pos_seg": seg}


def random_bert_weights(hidden=768, layers=12, heads=12, ffn=3072, vocab=30522, max_pos=512, type_vocab=2, seed=0, std=0.05):
    """Seeded stand-in for a checkpoint (there is no network for real ones).  Wider than HF's 0.02
    init and with non-trivial LayerNorm/bias terms so that every term of the forward matters.  HF state_dict names of a
    BertForSequenceClassification with these dimensions, values torch fp32."""
    import torch

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


def random_roberta_weights(hidden=768, layers=12, heads=12, ffn=3072, vocab=50265, max_pos=514, seed=0, std=0.05):
    """`random_bert_weights` under the HF state_dict names of a RobertaForSequenceClassification (one token type; the pooler / classifier
    tensors become the head's `classifier.dense` / `classifier.out_proj`)."""
    w = random_bert_weights(hidden, layers, heads, ffn, vocab, max_pos, type_vocab=1, seed=seed, std=std)
    out = {}
    for k, v in w.items():
        if k.startswith("bert.pooler.dense."):
            out["classifier.dense." + k.rsplit(".", 1)[1]] = v
        elif k.startswith("classifier."):
            out["classifier.out_proj." + k.rsplit(".", 1)[1]] = v
        else:
            out["roberta." + k[len("bert."):]] = v
    return out
