"""PACRR behind the reference plugin surface (capreolus/reranker/PACRR.py:81-117), scored by the fused gfx950 kernel in
capreolus_amd/csrc/pacrr.hip through the C ABI (a sibling model on the gather / similarity front end of
KNRM and DRMM).  Parameter names follow the reference state_dict (``ngrams.{i}.conv.*``, ``linear1..3.*`` - which the
reference also exposes as ``combine.{0,2,4}.*`` - and ``embedding.weight``)."""
import numpy as np
import torch
from torch import nn

from .. import engine
from . import Reranker


class _ConvMax2d(nn.Module):
    """Parameter holder with the reference's names (PACRRConvMax2dModule, PACRR.py:57-70); the arithmetic is in pacrr.hip."""

    def __init__(self, shape, n_filters, k):
        super().__init__()
        self.shape, self.k = shape, k
        self.conv = nn.Conv2d(1, n_filters, shape)


class PACRR_class(nn.Module):
    def __init__(self, extractor, config):
        super().__init__()
        p = dict(config)
        self.p = p
        if p["nonlinearity"] not in engine.NONLINEARITIES:
            raise ValueError("nonlinearity must be none, relu or tanh")
        weights = torch.as_tensor(np.asarray(extractor.embeddings, dtype=np.float32))
        self.embedding = nn.Embedding(*weights.shape)
        self.embedding.weight.data.copy_(weights)
        self.embedding.weight.requires_grad = False
        self.ngrams = nn.ModuleList(_ConvMax2d(ng, p["nfilters"], p["kmax"]) for ng in range(p["mingram"], p["maxgram"] + 1))
        qterm_size = len(self.ngrams) * p["kmax"] + (1 if p["idf"] else 0)
        self.linear1 = nn.Linear(extractor.config["maxqlen"] * qterm_size, p["combine"])
        self.linear2 = nn.Linear(p["combine"], p["combine"])
        self.linear3 = nn.Linear(p["combine"], 1)
        act = {"none": nn.Identity, "relu": nn.ReLU, "tanh": nn.Tanh}[p["nonlinearity"]]
        self.combine = nn.Sequential(self.linear1, act(), self.linear2, act(), self.linear3)   # same tensors, the reference's second set of names
        self._packed = engine.PackedEmbedding()

    def forward(self, doc, query, query_idf):
        if torch.is_grad_enabled() and self.training:
            return self._forward_train(doc, query, query_idf)
        w = self.embedding.weight
        conv_w = torch.cat([m.conv.weight.detach().reshape(-1) for m in self.ngrams]).contiguous()
        conv_b = torch.cat([m.conv.bias.detach().reshape(-1) for m in self.ngrams]).contiguous()
        p = self.p
        out = engine.pacrr_forward(query, doc, query_idf, self._packed.get(w), w.shape[0], w.shape[1], p["mingram"], p["maxgram"], p["nfilters"],
                                   p["kmax"], conv_w, conv_b, p["idf"], p["nonlinearity"], self.linear1.weight.detach().contiguous(),
                                   self.linear1.bias.detach(), self.linear2.weight.detach().contiguous(), self.linear2.bias.detach(),
                                   self.linear3.weight.detach().contiguous().view(-1), self.linear3.bias.detach())
        return out.view(-1, 1)

    def forward_lists(self, offsets, query=None, doc=None, idf=None, store=None, pair_q=None, pair_d=None):
        """Whole candidate lists through capamd_pacrr_forward_lists -> [B] (see KNRM_class.forward_lists).  Up to four query terms and
        32 filters (the MFMA kernel's geometry)."""
        w = self.embedding.weight
        conv_w = torch.cat([m.conv.weight.detach().reshape(-1) for m in self.ngrams]).contiguous()
        conv_b = torch.cat([m.conv.bias.detach().reshape(-1) for m in self.ngrams]).contiguous()
        p = self.p
        return engine.pacrr_forward_lists(
            offsets, store.idf_table if store is not None else idf, self._packed.get(w), w.shape[0], w.shape[1], p["mingram"], p["maxgram"], p["nfilters"],
            p["kmax"], conv_w, conv_b, p["idf"], p["nonlinearity"], self.linear1.weight.detach().contiguous(), self.linear1.bias.detach(),
            self.linear2.weight.detach().contiguous(), self.linear2.bias.detach(), self.linear3.weight.detach().contiguous().view(-1),
            self.linear3.bias.detach(), query=query, doc=doc, store=store, pair_q=pair_q, pair_d=pair_d)

    def _forward_train(self, doc, query, query_idf):
        """Training step (reference trainer/pytorch.py:96-99 -> PACRR.score), on HIP kernels end to end up to the three small linear
        layers: the [B, Q, L] similarity matrix (capamd_similarity_matrix; the table is frozen, no gradient flows through it), then
        the trainable stage of PACRR.py:68-78 - zero padding, the n-gram Conv2d, ReLU, max over filters, k-max over the document -
        forward with the winners' coordinates and backward into the convolution weights (pacrr_train.hip through
        `engine.PacrrConvMax`).  The idf softmax and `combine` (:46-54) run under autograd on the [B, Q, n] features.  Geometries
        beyond that kernel's limits keep the reference's op sequence (`_forward_train_aten`)."""
        import torch.nn.functional as F

        w = self.embedding.weight
        sim = engine.similarity_matrix(query, doc, self._packed.get(w), w.shape[0], w.shape[1])
        B, Q, L = sim.shape
        p = self.p
        if Q > 8 or L > 1024 or p["maxgram"] > 3 or p["kmax"] > 4 or p["nfilters"] > 256 or L < p["kmax"]:
            return self._forward_train_aten(doc, query, query_idf)
        conv_w = torch.cat([m.conv.weight.reshape(-1) for m in self.ngrams])
        conv_b = torch.cat([m.conv.bias.reshape(-1) for m in self.ngrams])
        feats = [engine.PacrrConvMax.apply(sim, conv_w, conv_b, p["mingram"], p["maxgram"], p["nfilters"], p["kmax"])]
        if p["idf"]:
            feats.append(F.softmax(query_idf.float(), dim=1).view(B, Q, 1))
        scores = torch.cat(feats, dim=2).reshape(B, -1)
        return self.combine(scores)

    def fused_train_step(self, d, optimizer, softmax=False):
        """One whole training step on the device (capamd_pacrr_train_step: similarity matrices, the convolution / k-max stage, the three
        Linear layers, the pairwise loss, backward, Adam - six launches, no autograd) - or None where the geometry is beyond its kernels
        (then the autograd route over the same HIP stages).  Parameters and Adam moments are updated in place; their version counters
        are bumped."""
        p = self.p
        B, Q, L = d["query"].shape[0], d["query"].shape[1], d["posdoc"].shape[1]
        if Q > 8 or L > 1024 or p["maxgram"] > 3 or p["kmax"] > 4 or p["nfilters"] > 256 or L < p["kmax"] or \
                not engine.pacrr_train_step_fits(B, Q, len(self.ngrams), p["kmax"], p["idf"], p["combine"]):
            return None
        params = []
        for m in self.ngrams:
            params += [m.conv.weight, m.conv.bias]
        params += [self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, self.linear3.weight, self.linear3.bias]
        hit = self.__dict__.get("_adam_step")
        if hit is None or hit.optimizer is not optimizer or hit.key[: len(params)] != tuple(t.data_ptr() for t in params) or not hit.still_valid():
            hit = self.__dict__["_adam_step"] = engine.AdamStep(optimizer, params)
        w = self.embedding.weight
        loss = engine.pacrr_train_step(d["query"], d["posdoc"], d["negdoc"], d["query_idf"], self._packed.get(w), w.shape[0], w.shape[1], p["mingram"],
                                       p["maxgram"], p["nfilters"], p["kmax"], p["idf"], p["combine"], p["nonlinearity"], hit, softmax)
        with torch.no_grad():
            torch._foreach_mul_(hit.trained, 1.0)          # (exact no-op: the kernels wrote the parameters behind autograd's back)
        return loss[0]

    def _forward_train_aten(self, doc, query, query_idf):
        """The trainable stage as the reference's ATen op sequence under autograd on the HIP similarity matrix: the checker of the
        HIP training path in the tests, and the route of geometries `engine.PacrrConvMax` does not take."""
        import torch.nn.functional as F

        w = self.embedding.weight
        sim = engine.similarity_matrix(query, doc, self._packed.get(w), w.shape[0], w.shape[1])
        B, Q, L = sim.shape
        x = sim.view(B, 1, Q, L)
        feats = []
        for m in self.ngrams:
            xp = F.pad(x, (0, m.shape - 1, 0, m.shape - 1)) if m.shape != 1 else x
            top_filters = F.relu(m.conv(xp)).max(dim=1)[0]
            feats.append(top_filters.topk(m.k, dim=2)[0])
        if self.p["idf"]:
            feats.append(F.softmax(query_idf.float(), dim=1).view(B, Q, 1))
        scores = torch.cat(feats, dim=2).reshape(B, -1)
        return self.combine(scores)


class PACRR(Reranker):
    """Hui, Yates, Berberich, de Melo. PACRR: A Position-Aware Neural IR Model for Relevance Matching. EMNLP 2017
    (reference PACRR.py:81-98)."""

    module_name = "PACRR"
    supports_resident = True   # term-id rows: served from a device-resident CandidateStore (Reranker.test_resident)
    config_spec = {"mingram": 1, "maxgram": 3, "nfilters": 32, "idf": True, "kmax": 2, "combine": 32, "nonlinearity": "relu"}

    def build_model(self):
        if not hasattr(self, "model"):
            self.model = PACRR_class(self.extractor, self.config)
        return self.model

    def score(self, d):
        return self._score_pos_neg(d)

    def test(self, d):
        return self.model(d["posdoc"], d["query"], d["query_idf"]).view(-1)

    def fused_train_step(self, d, optimizer, softmax=False):
        return self.model.fused_train_step(d, optimizer, softmax)

    def fused_step_available(self, batch_size):
        """whether `fused_train_step` takes this configuration (the training kernels' geometry, a batch whose activations fit one workgroup)"""
        c = self.config
        return c["maxgram"] <= 3 and c["kmax"] <= 4 and c["nfilters"] <= 256 and \
            engine.pacrr_train_step_fits(batch_size, self.extractor.config["maxqlen"], c["maxgram"] - c["mingram"] + 1, c["kmax"], c["idf"], c["combine"])

    lists_bit_identical = True # (the similarity matrix is the same: lookups of bit-identical similarities)

    @property
    def supports_lists(self):      # whole candidate lists (capamd_pacrr_forward_lists): the MFMA kernel's geometry only
        return self.model.p["nfilters"] <= 32 and self.model.p["maxgram"] <= 3

    def test_lists(self, d, offsets):
        return self.model.forward_lists(offsets, query=d["query"], doc=d["posdoc"], idf=d["query_idf"])

    def test_resident_lists(self, store, pair_q, pair_d, offsets):
        return self.model.forward_lists(offsets, store=store, pair_q=pair_q, pair_d=pair_d)
