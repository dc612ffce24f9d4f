"""DRMM-TKS behind the reference plugin surface (capreolus/reranker/DRMMTKS.py:68-110), scored by the fused gfx950
kernel in capreolus_amd/csrc/drmmtks.hip through the C ABI (a sibling model that reuses the
gather / similarity front end of KNRM and DRMM).  Parameter names follow the reference state_dict
(``ffw.0.*``, ``gates.weight``, ``output_layer.*``, ``embedding.weight``)."""
import numpy as np
import torch
from torch import nn

from .. import engine
from . import Reranker


class DRMMTKS_class(nn.Module):
    def __init__(self, extractor, config):
        super().__init__()
        self.topk = config["topk"]
        self.gate_type = config["gateType"]
        if self.gate_type != "IDF":
            raise NotImplementedError("DRMMTKS gateType=TV: the reference feeds integer ids to nn.Linear (DRMMTKS.py:42); only IDF is scored")
        weights = torch.as_tensor(np.asarray(extractor.embeddings, dtype=np.float32))
        self.embedding = nn.Embedding(*weights.shape)
        self.embedding.weight.data.copy_(weights)
        self.embedding.weight.requires_grad = not config["freezeemb"]          # create_emb_layer(non_trainable=freezeemb), DRMMTKS.py:25
        self.ffw = nn.Sequential(nn.Linear(self.topk, 1), nn.Tanh())
        self.gates = nn.Linear(1, 1, bias=False)
        self.output_layer = nn.Linear(1, 1)
        nn.init.uniform_(self.ffw[0].weight, -0.1, 0.1)   # MatchZoo-style initialisation (DRMMTKS.py:28-30)
        nn.init.uniform_(self.gates.weight, -0.01, 0.01)
        self._packed = engine.PackedEmbedding()

    def forward(self, doc, query, query_idf):
        if torch.is_grad_enabled() and self.training:
            return self._forward_train(doc, query, query_idf)
        w = self.embedding.weight
        out = engine.drmmtks_forward(query, doc, query_idf, self._packed.get(w), w.shape[0], w.shape[1], self.topk,
                                     self.gates.weight.detach().view(-1), self.ffw[0].weight.detach().contiguous().view(-1),
                                     self.ffw[0].bias.detach(), self.output_layer.weight.detach().view(-1), self.output_layer.bias.detach())
        return out.view(-1, 1)


    def forward_lists(self, offsets, query=None, doc=None, idf=None, store=None, pair_q=None, pair_d=None):
        """Whole candidate lists through capamd_drmmtks_forward_lists -> [B] (see KNRM_class.forward_lists); a list takes the idf row of
        its first pair."""
        w = self.embedding.weight
        return engine.drmmtks_forward_lists(
            offsets, store.idf_table if store is not None else idf, self._packed.get(w), w.shape[0], w.shape[1], self.topk,
            self.gates.weight.detach().view(-1), self.ffw[0].weight.detach().contiguous().view(-1), self.ffw[0].bias.detach(),
            self.output_layer.weight.detach().view(-1), self.output_layer.bias.detach(), query=query, doc=doc, store=store, pair_q=pair_q, pair_d=pair_d)

    def fused_train_step(self, d, optimizer, softmax=False):
        """One whole training step on the device (capamd_drmmtks_train_step); parameters and Adam moments are updated in place."""
        if d["query"].shape[0] > 1024 or self.embedding.weight.requires_grad:
            return None
        params = [self.ffw[0].weight, self.ffw[0].bias, self.gates.weight, self.output_layer.weight, self.output_layer.bias]
        hit = self.__dict__.get("_adam_step")
        if hit is None or hit.optimizer is not optimizer or hit.key[: len(params)] != tuple(p.data_ptr() for p in params) or not hit.still_valid():
            hit = self.__dict__["_adam_step"] = engine.AdamStep(optimizer, params)
        w = self.embedding.weight
        loss = engine.drmmtks_train_step(d["query"], d["posdoc"], d["negdoc"], d["query_idf"], self._packed.get(w), w.shape[0], w.shape[1], self.topk, hit,
                                         softmax)
        with torch.no_grad():
            torch._foreach_mul_(hit.trained, 1.0)          # (exact no-op: the kernel wrote the parameters behind autograd's back)
        return loss[0]

    def _forward_train(self, doc, query, query_idf):
        """Training step (reference trainer/pytorch.py:96-99 -> DRMMTKS.score): the gather / similarity / top-k - everything that
        touches the [B, Q, L] tensors - is the HIP kernel (capamd_drmmtks_features); the embedding table is frozen, so no
        gradient flows through the similarities, and the Linear(topk, 1)/tanh, the idf gate and the output layer
        (DRMMTKS.py:57-62) run under autograd on the [B, Q, topk] features."""
        w = self.embedding.weight
        if w.requires_grad:
            # freezeemb=False: the table trains too - the similarity matrix and the top-k as ATen ops under autograd on the GPU (dense table
            # gradient, like the reference's nn.Embedding); the one DRMM-TKS training configuration that is not on the HIP feature kernel
            if self.topk > doc.shape[1]:
                raise RuntimeError("selected index k out of range")
            topk = torch.topk(engine.similarity_matrix_autograd(self.embedding, query, doc), k=self.topk, dim=-1)[0]
        else:
            topk = engine.drmmtks_features(query, doc, self._packed.get(w), w.shape[0], w.shape[1], self.topk)
        ffw_vec = self.ffw(topk).squeeze(-1)                                            # (B, Q)
        gate = self.gates(query_idf.float()[:, :, None]).squeeze(-1) + (1 - (query != 0).float()) * -1e7
        wgt = torch.softmax(gate, dim=1)
        return self.output_layer((wgt * ffw_vec).sum(dim=-1, keepdim=True))


class DRMMTKS(Reranker):
    """Guo et al., CIKM'16, MatchZoo's top-k variant (reference DRMMTKS.py:68-80)."""

    module_name = "DRMMTKS"
    supports_resident = True   # term-id rows: served from a device-resident CandidateStore (Reranker.test_resident)
    config_spec = {"topk": 10, "gateType": "IDF", "freezeemb": True}

    def build_model(self):
        if not hasattr(self, "model"):
            self.model = DRMMTKS_class(self.extractor, self.config)
        return self.model

    def score(self, d):
        return self._score_pos_neg(d)

    def test(self, d):
        return self.model(d["posdoc"], d["query"], d["query_idf"]).view(-1)

    def fused_train_step(self, d, optimizer, softmax=False):
        return self.model.fused_train_step(d, optimizer, softmax)

    def fused_step_available(self, batch_size):
        return batch_size <= 1024 and bool(self.config["freezeemb"])

    supports_lists = True      # whole candidate lists: every distinct term of a list gathered once (capamd_drmmtks_forward_lists)
    lists_max_qlen = 8         # (two blocks of four query terms: csrc/lists.h kListMaxQ)
    lists_bit_identical = True # (top-k selections of bit-identical similarities, fed to the Linear in the same order)

    def test_lists(self, d, offsets):
        return self.model.forward_lists(offsets, query=d["query"], doc=d["posdoc"], idf=d["query_idf"])

    def test_resident_lists(self, store, pair_q, pair_d, offsets):
        return self.model.forward_lists(offsets, store=store, pair_q=pair_q, pair_d=pair_d)
