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
            raise NotImplementedError("the ConvKNRM training step is not part of the MI355X engine; score under model.eval()")
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


class ConvKNRM(Reranker):
    """Dai, Xiong, Callan, Liu. Convolutional Neural Networks for Soft-Matching N-Grams in Ad-hoc Search. WSDM'18
    (reference ConvKNRM.py:81-97)."""

    module_name = "ConvKNRM"
    config_spec = {"gradkernels": True, "maxngram": 3, "crossmatch": True, "filters": 128, "scoretanh": False, "singlefc": True}

    def build_model(self):
        if not hasattr(self, "model"):
            self.model = ConvKNRM_class(self.extractor, self.config)
        return self.model

    def score(self, d):
        q, idf = d["query"], d["query_idf"]
        return [self.model(d["posdoc"], q, idf).view(-1), self.model(d["negdoc"], q, idf).view(-1)]

    def test(self, d):
        return self.model(d["posdoc"], d["query"], d["query_idf"]).view(-1)

    def zero_grad(self, *args, **kwargs):
        self.model.zero_grad(*args, **kwargs)
