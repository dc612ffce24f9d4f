"""ConvKNRM behind the reference plugin surface (capreolus/reranker/ConvKNRM.py:81-120), scored by the fused gfx950 kernel in
capreolus_amd/csrc/convknrm.hip through the C ABI.

The module holds the parameters under the reference's state_dict names (``embeddings.weight``, ``kernels.kernels.{k}.mu|sigma``,
``convs.{g}.0.weight|bias``, ``combine.{0,2}.weight|bias``) so checkpoints interchange.  The convolutions never run at scoring
time: they are folded, once per set of weights, into per-token projection tables (engine.ConvProjectionTables).
"""
import numpy as np
import torch
from torch import nn

from .. import engine
from . import Reranker
from .KNRM import _MUS, _SIGMAS, _RbfBank


class ConvKNRM_class(nn.Module):
    def __init__(self, extractor, config):
        super().__init__()
        self.p = dict(config)
        weights = torch.as_tensor(np.asarray(extractor.embeddings, dtype=np.float32))
        self.embeddings = nn.Embedding(*weights.shape)
        self.embeddings.weight.data.copy_(weights)
        self.embeddings.weight.requires_grad = False          # create_emb_layer(non_trainable=True), ConvKNRM.py:17
        self.kernels = _RbfBank(_MUS, _SIGMAS, requires_grad=config["gradkernels"])
        G = config["maxngram"]
        self.convs = nn.ModuleList(nn.ModuleList([nn.Conv1d(weights.shape[1], config["filters"], g)]) for g in range(1, G + 1))
        channels = G * G if config["crossmatch"] else G
        K = self.kernels.count()
        steps = [nn.Linear(K * channels, 1)] if config["singlefc"] else [nn.Linear(K * channels, 30), nn.Tanh(), nn.Linear(30, 1)]
        if config["scoretanh"]:
            steps.append(nn.Tanh())
        self.combine = nn.Sequential(*steps)
        self._tables = engine.ConvProjectionTables()

    def forward(self, sentence, query_sentence, query_idf=None):
        """[B, 1] scores.  query_idf is accepted and ignored, as in the reference (ConvKNRM.py:42)."""
        if torch.is_grad_enabled() and self.training:
            return self._forward_train(sentence, query_sentence)
        w = self.embeddings.weight
        tables = self._tables.get(w, [c[0].weight for c in self.convs], [c[0].bias for c in self.convs])
        mu, sigma = self.kernels.stacked()
        lin1 = self.combine[0]
        w2 = b2 = None
        if not self.p["singlefc"]:
            w2, b2 = self.combine[2].weight.detach().contiguous().view(-1), self.combine[2].bias.detach()
        out = engine.convknrm_forward(query_sentence, sentence, tables, w.shape[0], self.p["maxngram"], self.p["filters"], self.p["crossmatch"],
                                      mu, sigma, lin1.weight.detach().contiguous(), lin1.bias.detach(), w2, b2, score_tanh=self.p["scoretanh"])
        return out.view(-1, 1)

    def forward_lists(self, offsets, query=None, doc=None, idf=None, store=None, pair_q=None, pair_d=None):
        """[B] scores of pairs laid out list after list (`offsets`: host array of n_lists + 1 boundaries; every list against its first
        pair's query): the unigram document view once per distinct token of a list (capamd_convknrm_forward_lists).  Equal to `forward`
        bit for bit.  Ids as [B, Q] / [B, L] rows, or index pairs of a device-resident CandidateStore (the id rows are then gathered on the
        device: int32 tables -> the int64 rows the kernels read)."""
        if store is not None:
            query_sentence, sentence = store.q_table.index_select(0, pair_q.long()).long(), store.d_table.index_select(0, pair_d.long()).long()
        else:
            query_sentence, sentence = query, doc
        w = self.embeddings.weight
        tables = self._tables.get(w, [c[0].weight for c in self.convs], [c[0].bias for c in self.convs])
        mu, sigma = self.kernels.stacked()
        lin1 = self.combine[0]
        w2 = b2 = None
        if not self.p["singlefc"]:
            w2, b2 = self.combine[2].weight.detach().contiguous().view(-1), self.combine[2].bias.detach()
        return engine.convknrm_forward_lists(offsets, query_sentence, sentence, tables, w.shape[0], self.p["maxngram"], self.p["filters"],
                                             self.p["crossmatch"], mu, sigma, lin1.weight.detach().contiguous(), lin1.bias.detach(), w2, b2,
                                             score_tanh=self.p["scoretanh"])


    def _forward_train(self, sentence, query_sentence):
        """Training step (reference trainer/pytorch.py:96-99 -> ConvKNRM.score).  The trainable convolutions sit IN FRONT of the
        similarity matrices, so the projection tables of the scoring kernel (rebuilt from the weights, not differentiable) cannot be
        used.  Two HIP stages, each one kernel forward and one backward: the n-gram convolutions over the frozen table as fp32
        matrix-pipe products that gather the table's rows themselves (capreolus_amd/csrc/ngram_conv.hip through `engine.NgramConv`:
        no embedding tensor, no permutes, no padded copies, no library convolution), and everything behind them - cosine similarity of
        every (query view, document view) pair, pad masks, RBF kernel pooling, log / mask / sum - in capreolus_amd/csrc/kernel_pool.hip
        (`engine.KernelPool`), which hands the gradient back to both convolution outputs and to the kernels' mu / sigma.  Geometries
        outside those kernels' limits (filters > 256 or not a multiple of 4, more than 24 query vectors per document view, an embedding
        width that is not a multiple of 4 or beyond 316, n-grams beyond 4) keep the reference's op sequence under autograd
        (`_forward_train_aten`)."""
        engine._need_gpu(sentence, query_sentence, self.embeddings.weight)
        G, Q = len(self.convs), query_sentence.shape[1]
        nf, D = self.p["filters"], self.embeddings.weight.shape[1]
        if nf % 4 or nf > 256 or (G if self.p["crossmatch"] else 1) * Q > 24 or self.kernels.count() > 16 or D % 4 or D > 316 or G > 4:
            return self._forward_train_aten(sentence, query_sentence)
        wb = []
        for conv in self.convs:
            wb += [conv[0].weight, conv[0].bias]
        a_reps, b_reps = engine.NgramConv.apply(query_sentence, sentence, self.embeddings.weight, *wb)      # [B, G, Q, F], [B, G, L, F]
        mu = torch.stack([k.mu for k in self.kernels.kernels]).float()          # live parameters: gradkernels trains them (ConvKNRM.py:22)
        sigma = torch.stack([k.sigma for k in self.kernels.kernels]).float()
        feats = engine.KernelPool.apply(a_reps, b_reps, query_sentence, sentence, mu, sigma, bool(self.p["crossmatch"]))
        return self.combine(feats)

    def fused_train_step(self, d, optimizer, softmax=False):
        """One whole training step on the device (capamd_convknrm_train_step: convolutions, kernel pooling, combine, pairwise loss,
        backward, Adam - eleven launches, no autograd) - or None when this configuration keeps the autograd route (a two-layer `combine`,
        geometries beyond the kernels' limits).  Parameters and Adam moments are updated in place; their version counters are bumped so
        that weight-derived caches (the scoring kernel's projection tables) notice."""
        G, Q = len(self.convs), d["query"].shape[1]
        nf, D = self.p["filters"], self.embeddings.weight.shape[1]
        if not self.p["singlefc"] or d["query"].shape[0] > 512:
            return None
        if nf % 4 or nf > 256 or (G if self.p["crossmatch"] else 1) * Q > 24 or self.kernels.count() > 16 or D % 4 or D > 316 or G > 4:
            return None
        ks = list(self.kernels.kernels)
        lin = self.combine[0]
        params = [k.mu for k in ks] + [k.sigma for k in ks]
        for conv in self.convs:
            params += [conv[0].weight, conv[0].bias]
        params += [lin.weight, lin.bias]
        hit = self.__dict__.get("_adam_step")
        if hit is None or hit.optimizer is not optimizer or hit.key[: len(params)] != tuple(p.data_ptr() for p in params) or not hit.still_valid():
            hit = self.__dict__["_adam_step"] = engine.AdamStep(optimizer, params)
        loss = engine.convknrm_train_step(d["query"], d["posdoc"], d["negdoc"], self.embeddings.weight, G, nf, len(ks), bool(self.p["crossmatch"]), hit,
                                          bool(self.p["scoretanh"]), softmax)
        with torch.no_grad():
            torch._foreach_mul_(hit.trained, 1.0)          # (exact no-op: the kernels wrote the parameters behind autograd's back)
        return loss[0]

    def _forward_train_aten(self, sentence, query_sentence):
        """The reference's arithmetic (ConvKNRM.py:42-77, common.py:195-221) as ATen ops under autograd: the checker of the HIP
        training path in the tests, and the route of geometries the kernel-pooling kernel does not take."""
        import torch.nn.functional as F

        engine._need_gpu(sentence, query_sentence, self.embeddings.weight)
        a_emb, b_emb = self.embeddings(query_sentence).permute(0, 2, 1), self.embeddings(sentence).permute(0, 2, 1)
        a_reps, b_reps = [], []
        for g, conv in enumerate(self.convs, start=1):
            a_reps.append(conv[0](F.pad(a_emb, (0, g - 1))).permute(0, 2, 1))
            b_reps.append(conv[0](F.pad(b_emb, (0, g - 1))).permute(0, 2, 1))
        pairs = [(a, b) for a in a_reps for b in b_reps] if self.p["crossmatch"] else list(zip(a_reps, b_reps))
        q_pad, d_pad = (query_sentence == 0)[:, :, None], (sentence == 0)[:, None, :]      # extractor.pad == 0
        sims = []
        for a, b in pairs:
            den = (a.norm(p=2, dim=2)[:, :, None] + 1e-9) * (b.norm(p=2, dim=2)[:, None, :] + 1e-9)
            sim = a.bmm(b.permute(0, 2, 1)) / den
            sims.append(sim.masked_fill(q_pad, 0.0).masked_fill(d_pad, 0.0))
        simmats = torch.stack(sims, dim=1)                                   # [B, VIEWS, Q, L]
        mu = torch.stack([k.mu for k in self.kernels.kernels]).float()
        sigma = torch.stack([k.sigma for k in self.kernels.kernels]).float()
        adj = simmats[:, None] - mu.view(1, -1, 1, 1, 1)
        kernels = torch.exp(-0.5 * adj * adj / sigma.view(1, -1, 1, 1, 1) / sigma.view(1, -1, 1, 1, 1))   # [B, K, VIEWS, Q, L]
        B, K, V, Q, L = kernels.shape
        result = kernels.reshape(B, K * V, Q, L).sum(dim=3)
        mask = (simmats.sum(dim=3) != 0.0)[:, None].expand(B, K, V, Q).reshape(B, K * V, Q)
        result = torch.where(mask, (result + 1e-6).log(), mask.float()).sum(dim=2)
        return self.combine(result)


class ConvKNRM(Reranker):
    """Dai, Xiong, Callan, Liu. Convolutional Neural Networks for Soft-Matching N-Grams in Ad-hoc Search. WSDM'18
    (reference ConvKNRM.py:81-97)."""

    module_name = "ConvKNRM"
    supports_resident = True   # term-id rows: served from a device-resident CandidateStore (Reranker.test_resident)
    config_spec = {"gradkernels": True, "maxngram": 3, "crossmatch": True, "filters": 128, "scoretanh": False, "singlefc": True}

    def build_model(self):
        if not hasattr(self, "model"):
            self.model = ConvKNRM_class(self.extractor, self.config)
        return self.model

    def score(self, d):
        return self._score_pos_neg(d)

    def test(self, d):
        return self.model(d["posdoc"], d["query"], d["query_idf"]).view(-1)

    # Whole candidate lists (capamd_convknrm_forward_lists: the unigram document view once per distinct token of a list) exist, are
    # bit-identical to the per-pair kernel (the same matrix instructions on the same operands; the n-gram views and the pooling are
    # the per-pair kernel's) and are NOT what `PytorchTrainer.predict` takes: measured 9.53 M pairs/s against 9.70 M pair by pair on the
    # benchmark's lists (profiles/r06/convknrm_lists.txt) - the per-pair kernel is bound by its gather, of which the unigram part is a
    # sixth, and the list's own passes (mark 0.11 ms, the table 0.30 ms) cost what the shorter gather saves (0.45 of 6.6 ms).
    # `test_lists` / `test_resident_lists` stay callable for callers that want them.
    supports_lists = False
    lists_bit_identical = True
    lists_max_qlen = 8

    def test_lists(self, d, offsets):
        return self.model.forward_lists(offsets, query=d["query"], doc=d["posdoc"])

    def test_resident_lists(self, store, pair_q, pair_d, offsets):
        return self.model.forward_lists(offsets, store=store, pair_q=pair_q, pair_d=pair_d)

    def fused_train_step(self, d, optimizer, softmax=False):
        return self.model.fused_train_step(d, optimizer, softmax)

    def fused_step_available(self, batch_size):
        """whether `fused_train_step` takes this configuration: every limit that method checks per batch (a single-Linear `combine`,
        <= 512 pairs, the HIP training kernels' geometry - filters, n-gram sizes x query terms, kernel bank, embedding width), from the
        configuration, the extractor's `maxqlen` and the built model, so that the trainer chooses between the device step and the graph
        route BEFORE it builds its optimizer"""
        c = self.config
        if not (bool(c["singlefc"]) and batch_size <= 512 and c["filters"] % 4 == 0 and c["filters"] <= 256 and c["maxngram"] <= 4):
            return False
        cfg = getattr(getattr(self, "extractor", None), "config", None)
        maxqlen = int(cfg["maxqlen"]) if isinstance(cfg, dict) and "maxqlen" in cfg else 4
        if (c["maxngram"] if c["crossmatch"] else 1) * maxqlen > 24:
            return False
        m = getattr(self, "model", None)
        if m is not None:
            D = m.embeddings.weight.shape[1]
            if m.kernels.count() > 16 or D % 4 or D > 316:
                return False
        return True

    def zero_grad(self, *args, **kwargs):
        self.model.zero_grad(*args, **kwargs)
