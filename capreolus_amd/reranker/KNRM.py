"""KNRM behind the reference plugin surface (capreolus/reranker/KNRM.py:58-101), scored by the
fused gfx950 kernel (capreolus_amd/csrc/knrm.hip) through the C ABI.

The nn.Module only *holds* parameters, under the reference's state_dict names
(``kernels.kernels.{k}.mu|sigma``, ``embedding.weight``, ``combine.{0,2}.weight|bias`` —
SURVEY.md §8b) so checkpoints interchange; its forward is one C-ABI call.
"""
import numpy as np
import torch
from torch import nn

from .. import engine
from . import Reranker

_MUS = [-0.9, -0.7, -0.5, -0.3, -0.1, 0.1, 0.3, 0.5, 0.7, 0.9, 1.0]  # reference KNRM.py:20
_SIGMAS = [0.1] * 10 + [0.001]                                        # reference KNRM.py:21


class _Rbf(nn.Module):
    def __init__(self, mu, sigma, requires_grad):
        super().__init__()
        self.mu = nn.Parameter(torch.tensor(mu), requires_grad=requires_grad)
        self.sigma = nn.Parameter(torch.tensor(sigma), requires_grad=requires_grad)


class _RbfBank(nn.Module):
    def __init__(self, mus, sigmas, requires_grad):
        super().__init__()
        self.kernels = nn.ModuleList([_Rbf(m, s, requires_grad) for m, s in zip(mus, sigmas)])

    def count(self):
        return len(self.kernels)

    def stacked(self):
        """(mu[K], sigma[K]) as contiguous device vectors, re-read from the live parameters (one scalar nn.Parameter per kernel and
        quantity, as the reference's state_dict names them: common.py:229-230) - kept until a parameter changes (storage or version
        counter; PytorchTrainer bumps the counters after graph replays, which update parameters behind autograd's back)."""
        key = tuple((p.data_ptr(), p._version) for k in self.kernels for p in (k.mu, k.sigma))
        hit = self.__dict__.get("_stacked")
        if hit is None or hit[0] != key or torch.is_grad_enabled():
            mu = torch.stack([k.mu.detach() for k in self.kernels]).float()
            sigma = torch.stack([k.sigma.detach() for k in self.kernels]).float()
            hit = self.__dict__["_stacked"] = (key, mu, sigma)
        return hit[1], hit[2]


class KNRM_class(nn.Module):
    def __init__(self, extractor, config):
        super().__init__()
        self.p = config
        self.kernels = _RbfBank(_MUS, _SIGMAS, requires_grad=config["gradkernels"])
        weights = torch.as_tensor(np.asarray(extractor.embeddings, dtype=np.float32))
        self.embedding = nn.Embedding(*weights.shape)
        self.embedding.weight.data.copy_(weights)
        self.embedding.weight.requires_grad = bool(config["finetune"])
        K = self.kernels.count()
        steps = [nn.Linear(K, 1)] if config["singlefc"] else [nn.Linear(K, 30), nn.Tanh(), nn.Linear(30, 1)]
        if config["scoretanh"]:
            steps.append(nn.Tanh())
        self.combine = nn.Sequential(*steps)
        self._packed = engine.PackedEmbedding()

    def forward(self, doctoks, querytoks, query_idf=None):
        """[B, 1] scores.  query_idf is accepted and ignored, as in the reference (KNRM.py:39)."""
        if torch.is_grad_enabled() and self.training:
            return self._forward_train(doctoks, querytoks)
        w = self.embedding.weight
        packed = self._packed.get(w)
        mu, sigma = self.kernels.stacked()
        lin1 = self.combine[0]
        if self.p["singlefc"]:
            w2 = b2 = None
        else:
            w2, b2 = self.combine[2].weight.detach(), self.combine[2].bias.detach()
        out = engine.knrm_forward(
            querytoks, doctoks, packed, w.shape[0], w.shape[1], mu, sigma, lin1.weight.detach().contiguous(),
            lin1.bias.detach(), w2, b2, scoretanh=self.p["scoretanh"])
        return out.view(-1, 1)

    def _forward_train(self, doctoks, querytoks):
        """Training step (reference trainer/pytorch.py:96-99 -> KNRM.score): the gather / interaction / kernel pooling --
        everything that touches the [B, Q, L] tensors -- is the HIP kernel (capamd_knrm_features, which also returns
        d f/d mu and d f/d sigma); the 11 -> 1 `combine` and the loss stay under autograd on the [B, 11] features."""
        if self.embedding.weight.requires_grad:
            return self._forward_train_finetune(doctoks, querytoks)
        w = self.embedding.weight
        packed = self._packed.get(w)
        mu = torch.stack([k.mu for k in self.kernels.kernels]).float()
        sigma = torch.stack([k.sigma for k in self.kernels.kernels]).float()
        feats = engine.KnrmFeatures.apply(mu, sigma, querytoks, doctoks, packed, w.shape[0], w.shape[1])
        return self.combine(feats)

    def _forward_train_finetune(self, doctoks, querytoks):
        """`finetune=True` (KNRM.py:23: the embedding table trains too): the gradient has to reach the [V, D] table through both operands
        of every cosine, which the feature kernel does not provide - the reference's op sequence (SimilarityMatrix common.py:155-182,
        RbfKernelBank :224-250, KNRM.py:39-55) as ATen ops under autograd ON THE GPU, dense table gradient like the reference's
        nn.Embedding.  A rarely used option (the reference's own comment: "TODO check save when True"); scoring in eval mode stays on
        the fused kernels, whose packed copy of the table follows its version counter."""
        simmat = engine.similarity_matrix_autograd(self.embedding, querytoks, doctoks)                # [B, Q, L]
        mu = torch.stack([k.mu for k in self.kernels.kernels]).float().view(1, -1, 1, 1)
        sigma = torch.stack([k.sigma for k in self.kernels.kernels]).float().view(1, -1, 1, 1)
        adj = simmat[:, None] - mu
        kernels = torch.exp(-0.5 * adj * adj / sigma / sigma)                 # [B, K, Q, L]
        result = kernels.sum(dim=3)
        mask = (simmat.sum(dim=2) != 0.0)[:, None, :].expand_as(result)
        result = torch.where(mask, (result + 1e-6).log(), mask.float()).sum(dim=2)
        return self.combine(result)

    def fused_train_step(self, d, optimizer, softmax=False):
        """One whole training step on the device (capamd_knrm_train_step: score(pos), score(neg), pairwise loss, backward, Adam) - or
        None when this configuration keeps the autograd route (a two-layer `combine`, `finetune`).  Parameters and Adam moments are
        updated in place; their version counters are bumped so that weight-derived caches notice."""
        if not self.p["singlefc"] or self.embedding.weight.requires_grad or d["query"].shape[0] > 1024:
            return None
        ks = list(self.kernels.kernels)
        lin = self.combine[0]
        params = [k.mu for k in ks] + [k.sigma for k in ks] + [lin.weight, lin.bias]
        hit = self.__dict__.get("_adam_step")
        if hit is None or hit.optimizer is not optimizer or hit.key[: len(params)] != tuple(p.data_ptr() for p in params) or not hit.still_valid():
            hit = self.__dict__["_adam_step"] = engine.AdamStep(optimizer, params)
        w = self.embedding.weight
        loss = engine.knrm_train_step(d["query"], d["posdoc"], d["negdoc"], self._packed.get(w), w.shape[0], w.shape[1], len(ks), hit,
                                      bool(self.p["gradkernels"]), bool(self.p["scoretanh"]), softmax)
        with torch.no_grad():
            torch._foreach_mul_(hit.trained, 1.0)          # (exact no-op: the kernel wrote the parameters behind autograd's back)
        return loss[0]

    def forward_indexed(self, store, pair_q, pair_d):
        """Scores (query row, document row) pairs of a device-resident `CandidateStore` -> [B]."""
        w = self.embedding.weight
        packed = self._packed.get(w)
        mu, sigma = self.kernels.stacked()
        lin1 = self.combine[0]
        w2 = b2 = None
        if not self.p["singlefc"]:
            w2, b2 = self.combine[2].weight.detach(), self.combine[2].bias.detach()
        return engine.knrm_forward_indexed(store.q_table, store.d_table, pair_q, pair_d, packed, w.shape[0], w.shape[1], mu, sigma,
                                           lin1.weight.detach().contiguous(), lin1.bias.detach(), w2, b2, scoretanh=self.p["scoretanh"])


    def forward_lists(self, offsets, query=None, doc=None, store=None, pair_q=None, pair_d=None):
        """Whole candidate lists (pairs laid out list after list, `offsets` their boundaries on the host; every list scored against
        its first pair's query) through capamd_knrm_forward_lists -> [B]."""
        w = self.embedding.weight
        mu, sigma = self.kernels.stacked()
        lin1 = self.combine[0]
        w2 = b2 = None
        if not self.p["singlefc"]:
            w2, b2 = self.combine[2].weight.detach(), self.combine[2].bias.detach()
        return engine.knrm_forward_lists(offsets, self._packed.get(w), w.shape[0], w.shape[1], mu, sigma, lin1.weight.detach().contiguous(),
                                         lin1.bias.detach(), w2, b2, scoretanh=self.p["scoretanh"], query=query, doc=doc, store=store,
                                         pair_q=pair_q, pair_d=pair_d)


class KNRM(Reranker):
    """Xiong et al., End-to-End Neural Ad-hoc Ranking with Kernel Pooling, SIGIR'17 (reference KNRM.py:58-69)."""

    module_name = "KNRM"
    supports_resident = True   # term-id rows: served from a device-resident CandidateStore (Reranker.test_resident)
    config_spec = {"gradkernels": True, "scoretanh": False, "singlefc": True, "finetune": False}

    def build_model(self):
        if not hasattr(self, "model"):
            self.model = KNRM_class(self.extractor, self.config)
        return self.model

    def score(self, d):
        return self._score_pos_neg(d)

    def test(self, d):
        return self.model(d["posdoc"], d["query"], d["query_idf"]).view(-1)

    def fused_train_step(self, d, optimizer, softmax=False):
        return self.model.fused_train_step(d, optimizer, softmax)

    def fused_step_available(self, batch_size):
        """whether `fused_train_step` takes this configuration (a single-Linear `combine`, frozen table, batches of <= 1024 pairs)"""
        return bool(self.config["singlefc"]) and not self.config["finetune"] and batch_size <= 1024

    def test_resident(self, store, pair_q, pair_d):
        return self.model.forward_indexed(store, pair_q, pair_d)

    supports_lists = True      # whole candidate lists: every distinct term of a list gathered once (capamd_knrm_forward_lists)
    lists_max_qlen = 8         # (two blocks of four query terms: csrc/lists.h kListMaxQ)

    def test_lists(self, d, offsets):
        return self.model.forward_lists(offsets, query=d["query"], doc=d["posdoc"])

    def test_resident_lists(self, store, pair_q, pair_d, offsets):
        return self.model.forward_lists(offsets, store=store, pair_q=pair_q, pair_d=pair_d)
