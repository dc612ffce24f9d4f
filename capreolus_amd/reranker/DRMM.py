"""DRMM behind the reference plugin surface (capreolus/reranker/DRMM.py:119-155), scored by the
fused gfx950 kernel (capreolus_amd/csrc/drmm.hip) through the C ABI.

Parameter names follow the reference state_dict (``ffw.{0,2}.*``, ``gates.weight``,
``output_layer.*``, ``embedding.weight``).
"""
import numpy as np
import torch
from torch import nn

from .. import engine
from . import Reranker


class DRMM_class(nn.Module):
    def __init__(self, extractor, config):
        super().__init__()
        self.nbins = config["nbins"]
        self.nodes = config["nodes"]
        self.hist_type = config["histType"]
        self.gate_type = config["gateType"]
        if self.hist_type not in engine.HIST_TYPES:
            raise ValueError("Invalid value for histType: histType should be 'CH', 'NH', or 'LCH'")
        weights = torch.as_tensor(np.asarray(extractor.embeddings, dtype=np.float32))
        self.embedding = nn.Embedding(*weights.shape)
        self.embedding.weight.data.copy_(weights)
        self.embedding.weight.requires_grad = False
        self.ffw = nn.Sequential(nn.Linear(self.nbins + 1, self.nodes), nn.Tanh(), nn.Linear(self.nodes, 1), nn.Tanh())
        if self.gate_type == "IDF":
            self.gates = nn.Linear(1, 1, bias=False)
        elif self.gate_type == "TV":
            self.gates = nn.Linear(weights.shape[1], 1, bias=False)
        else:
            raise ValueError("Invalid value for gateType: gateType should be either IDF or TV")
        self.output_layer = nn.Linear(1, 1)
        # MatchZoo-style initialisation, as the reference (DRMM.py:36-39)
        nn.init.uniform_(self.ffw[0].weight, -0.1, 0.1)
        nn.init.uniform_(self.ffw[2].weight, -0.1, 0.1)
        nn.init.uniform_(self.gates.weight, -0.01, 0.01)
        self._packed = engine.PackedEmbedding()
        self._edges = None

    def _bin_edges(self, device):
        # same values the reference computes on the host: torch.linspace(-1, 1, nbins+1)[1:] (DRMM.py:63)
        if self._edges is None or self._edges.device != device or self._edges.numel() != self.nbins:
            self._edges = torch.linspace(-1, 1, self.nbins + 1)[1:].contiguous().to(device)
        return self._edges

    def forward(self, sentence, query_sentence, query_idf, counts_out=None):
        if torch.is_grad_enabled() and self.training:
            return self._forward_train(sentence, query_sentence, query_idf)
        w = self.embedding.weight
        packed = self._packed.get(w)
        out = engine.drmm_forward(
            query_sentence, sentence, query_idf, packed, w.shape[0], w.shape[1], self._bin_edges(w.device), self.hist_type,
            self.gate_type, self.gates.weight.detach().contiguous().view(-1), w.detach(), self.ffw[0].weight.detach().contiguous(),
            self.ffw[0].bias.detach(), self.ffw[2].weight.detach().contiguous().view(-1), self.ffw[2].bias.detach(),
            self.output_layer.weight.detach().view(-1), self.output_layer.bias.detach(), counts_out=counts_out)
        return out.view(-1, 1)

    def fused_train_step(self, d, optimizer, softmax=False):
        """One whole training step on the device (capamd_drmm_train_step) - or None when this configuration keeps the autograd route (the
        term-vector gate, more than 16 hidden nodes).  Parameters and Adam moments are updated in place."""
        if self.gate_type != "IDF" or self.nodes > 16 or 2 * d["query"].numel() > 1024:       # (the step kernel's limits: 2 B Q <= 1024)
            return None
        params = [self.ffw[0].weight, self.ffw[0].bias, self.ffw[2].weight, self.ffw[2].bias, self.gates.weight, self.output_layer.weight,
                  self.output_layer.bias]
        hit = self.__dict__.get("_adam_step")
        if hit is None or hit.optimizer is not optimizer or hit.key[: len(params)] != tuple(p.data_ptr() for p in params) or not hit.still_valid():
            hit = self.__dict__["_adam_step"] = engine.AdamStep(optimizer, params)
        w = self.embedding.weight
        loss = engine.drmm_train_step(d["query"], d["posdoc"], d["negdoc"], d["query_idf"], self._packed.get(w), w.shape[0], w.shape[1],
                                      self._bin_edges(w.device), self.hist_type, self.nodes, hit, softmax)
        with torch.no_grad():
            torch._foreach_mul_(hit.trained, 1.0)          # (exact no-op: the kernel wrote the parameters behind autograd's back)
        return loss[0]

    def _forward_train(self, sentence, query_sentence, query_idf):
        """Training step: the matching histogram (everything that touches [B, Q, L]) is the HIP kernel and has no
        trainable inputs (DRMM.py:22 freezes the embedding); the 30 -> 5 -> 1 net, the gate and the output layer --
        a few hundred flops per pair -- stay under autograd (DRMM.py:106-114)."""
        w = self.embedding.weight
        # (inside a HIP graph capture - PytorchTrainer's captured training step - a device -> host read is not allowed: there the
        # kernel's status word carries the same error, raised when the trainer reads it at the end of the iteration)
        if not torch.cuda.is_current_stream_capturing() and (query_sentence < 0).any():
            raise IndexError("index out of range in self: DRMM cannot score an OOV (negative) query term id")
        feats = engine.drmm_features(query_sentence, sentence, self._packed.get(w), w.shape[0], w.shape[1], self._bin_edges(w.device),
                                     self.hist_type)
        z = self.ffw(feats).squeeze(-1)                                    # [B, Q]
        qmask = (query_sentence != 0).float()
        if self.gate_type == "IDF":
            gl = self.gates(query_idf.float().unsqueeze(-1)).squeeze(-1)
        else:
            gl = self.gates(w[query_sentence.clamp(min=0)]).squeeze(-1)
        g = torch.softmax(gl + (1 - qmask) * -1e7, dim=1)
        return self.output_layer((g * z).sum(dim=-1, keepdim=True))

    def forward_indexed(self, store, pair_q, pair_d):
        """Scores (query row, document row) pairs of a device-resident `CandidateStore` -> [B]."""
        w = self.embedding.weight
        packed = self._packed.get(w)
        return engine.drmm_forward_indexed(
            store.q_table, store.d_table, store.idf_table, pair_q, pair_d, packed, w.shape[0], w.shape[1], self._bin_edges(w.device),
            self.hist_type, self.gate_type, self.gates.weight.detach().contiguous().view(-1), w.detach(),
            self.ffw[0].weight.detach().contiguous(), self.ffw[0].bias.detach(), self.ffw[2].weight.detach().contiguous().view(-1),
            self.ffw[2].bias.detach(), self.output_layer.weight.detach().view(-1), self.output_layer.bias.detach())


    def forward_lists(self, offsets, query=None, doc=None, idf=None, store=None, pair_q=None, pair_d=None):
        """Whole candidate lists through capamd_drmm_forward_lists -> [B] (see KNRM_class.forward_lists); a list takes the idf row of
        its first pair."""
        w = self.embedding.weight
        return engine.drmm_forward_lists(
            offsets, store.idf_table if store is not None else idf, self._packed.get(w), w.shape[0], w.shape[1], self._bin_edges(w.device), self.hist_type,
            self.gate_type, self.gates.weight.detach().contiguous().view(-1), w.detach(), self.ffw[0].weight.detach().contiguous(),
            self.ffw[0].bias.detach(), self.ffw[2].weight.detach().contiguous().view(-1), self.ffw[2].bias.detach(),
            self.output_layer.weight.detach().view(-1), self.output_layer.bias.detach(), query=query, doc=doc, store=store, pair_q=pair_q, pair_d=pair_d)


class DRMM(Reranker):
    """Guo et al., A Deep Relevance Matching Model for Ad-hoc Retrieval, CIKM'16 (reference DRMM.py:119-133)."""

    module_name = "DRMM"
    supports_resident = True   # term-id rows: served from a device-resident CandidateStore (Reranker.test_resident)
    config_spec = {"nbins": 29, "nodes": 5, "histType": "LCH", "gateType": "IDF"}

    def build_model(self):
        if not hasattr(self, "model"):
            self.model = DRMM_class(self.extractor, self.config)
        return self.model

    def score(self, d):
        return self._score_pos_neg(d)

    def test(self, d):
        return self.model(d["posdoc"], d["query"], d["query_idf"]).view(-1)

    def fused_train_step(self, d, optimizer, softmax=False):
        return self.model.fused_train_step(d, optimizer, softmax)

    def fused_step_available(self, batch_size):
        """whether `fused_train_step` takes this configuration (idf gate, <= 16 hidden nodes, 2 B Q <= 1024: B <= 128 at the extractor's
        default of four query terms - a longer `maxqlen` lowers the batch the device step takes, and the trainer must know BEFORE it builds
        its optimizer: ADVICE r4)"""
        cfg = getattr(getattr(self, "extractor", None), "config", None)
        maxqlen = int(cfg["maxqlen"]) if isinstance(cfg, dict) and "maxqlen" in cfg else 4
        return self.config["gateType"] == "IDF" and self.config["nodes"] <= 16 and 2 * batch_size * maxqlen <= 1024

    def test_resident(self, store, pair_q, pair_d):
        return self.model.forward_indexed(store, pair_q, pair_d)

    supports_lists = True      # whole candidate lists: every distinct term of a list gathered once (capamd_drmm_forward_lists)
    lists_max_qlen = 8         # (two blocks of four query terms: csrc/lists.h kListMaxQ)
    lists_bit_identical = True # (integer bin counts of bit-identical similarities)

    def test_lists(self, d, offsets):
        return self.model.forward_lists(offsets, query=d["query"], doc=d["posdoc"], idf=d["query_idf"])

    def test_resident_lists(self, store, pair_q, pair_d, offsets):
        return self.model.forward_lists(offsets, store=store, pair_q=pair_q, pair_d=pair_d)
