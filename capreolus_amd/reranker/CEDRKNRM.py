"""CEDR-KNRM behind the reference plugin surface (capreolus/reranker/CEDRKNRM.py:188-217), scored by the gfx950 BERT encoder of
ptBERTMaxP plus the kernels in capreolus_amd/csrc/cedr_tap.h / cedr.hip through the C ABI.

The module holds the parameters under the reference's state_dict names (``bert.embeddings.*``, ``bert.encoder.layer.N.*``,
``bert.pooler.dense.*``, ``kernels.kernels.{k}.mu|sigma``, ``combine.{0,1}.weight|bias``, ``one``, ``zero``) so checkpoints
interchange.  BERT-architecture encoders: BertModel checkpoints and ElectraModel checkpoints whose embedding size equals the
hidden size (electra-base: the same encoder without a pooler), or an explicit geometry.
"""
import torch
from torch import nn

from .. import engine
from . import Reranker
from .KNRM import _RbfBank
from .ptBERTMaxP import bert_body


class CEDRKNRM_Class(nn.Module):
    def __init__(self, extractor, config):
        super().__init__()
        self.extractor = extractor
        self.config = dict(config)
        pre = config["pretrained"]
        if isinstance(pre, dict):            # explicit geometry, weights loaded later with load_state_dict
            self.bert = bert_body(**pre)
        else:
            self.bert = self._from_hf(pre, config["hidden_dropout_prob"])
        self.hidden_size = self.bert.embeddings.word_embeddings.weight.shape[1]
        mus = list(config["mus"]) + [1.0]
        sigmas = [config["sigma"] for _ in config["mus"]] + [0.01]
        self.kernels = _RbfBank(mus, sigmas, requires_grad=config["gradkernels"])
        layers = [int(x) for x in config["simmat_layers"]]
        if -1 in layers:
            if len(layers) != 1 or config["cls"] is None:
                raise AssertionError("simmat_layers = [-1] needs cls to be set")   # CEDRKNRM.py:48-50
            self._layers, combine_size = [], 0
        else:
            self._layers, combine_size = layers, self.kernels.count() * len(layers)
        if config["cls"] not in ("avg", "max", None):
            raise AssertionError("cls must be avg, max or None")
        if config["cls"]:
            combine_size += self.hidden_size
        if config["combine_hidden"] == 0:
            steps = [nn.Linear(combine_size, 1)]
        else:
            steps = [nn.Linear(combine_size, config["combine_hidden"]), nn.Linear(config["combine_hidden"], 1)]
        for lin in steps:   # "weight init from PyTorch 0.4" (CEDRKNRM.py:61-72)
            lin.weight.data.uniform_(-1.0 / lin.weight.size(1) ** 0.5, 1.0 / lin.weight.size(1) ** 0.5)
        self.combine = nn.Sequential(*steps)
        self.num_passages = extractor.config["numpassages"]
        self.maxseqlen = extractor.config["maxseqlen"]
        self.maxqlen = extractor.config["maxqlen"] + 1       # incl. [SEP] (CEDRKNRM.py:77-79)
        self.one = nn.Parameter(torch.ones(1), requires_grad=False)
        self.zero = nn.Parameter(torch.zeros(1), requires_grad=False)
        self._engine = None
        self._dummy = None

    @staticmethod
    def _from_hf(name, hidden_dropout_prob):
        """Reads a local/cached HF checkpoint (no network here) and copies its tensors into the container."""
        from transformers import AutoModel

        name = {"bert-base-msmarco": "Capreolus/bert-base-msmarco", "electra-base-msmarco": "Capreolus/electra-base-msmarco",
                "electra-base": "google/electra-base-discriminator"}.get(name, name)      # the reference's aliases (CEDRKNRM.py:20-35)
        hf = AutoModel.from_pretrained(name, hidden_dropout_prob=hidden_dropout_prob)
        c = hf.config
        # BERT, or ELECTRA whose embedding size equals its hidden size (no embeddings_project): the same encoder arithmetic
        if c.model_type not in ("bert", "electra") or c.hidden_act != "gelu" or getattr(c, "embedding_size", c.hidden_size) != c.hidden_size:
            raise NotImplementedError(f"{name}: unsupported architecture {c.model_type}/{c.hidden_act}")
        body = bert_body(c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.intermediate_size, c.vocab_size,
                         c.max_position_embeddings, c.type_vocab_size, pooler=c.model_type == "bert")
        missing = body.load_state_dict(hf.state_dict(), strict=False)
        if missing.missing_keys:
            raise RuntimeError(f"checkpoint lacks {missing.missing_keys}")
        return body

    def _params(self):
        p = {"bert." + k: v for k, v in self.bert.state_dict(keep_vars=True).items() if not k.endswith("position_ids")}
        return p

    def forward(self, bert_input, bert_mask, bert_segments):
        if self.training:
            raise NotImplementedError("capreolus_amd scores with hand-written inference kernels; call under model.eval()")
        B = bert_input.shape[0]
        shape = (B, self.num_passages, self.maxseqlen)
        if self._engine is None:
            be = engine.BertEngine(self._params(), self.bert.num_attention_heads, microbatch=int(self.config.get("microbatch", 256)),
                                   compute_dtype=self.config.get("compute_dtype", "fp16"), skip_padding=bool(self.config.get("skip_padding", True)))
            self._engine = engine.CedrEngine(be)
        else:
            self._engine.be.params = self._params()
        mu, sigma = self.kernels.stacked()
        lin1 = self.combine[0]
        w2 = b2 = None
        if len(self.combine) == 2:
            w2, b2 = self.combine[1].weight.detach().contiguous().view(-1), self.combine[1].bias.detach()
        out = self._engine.forward(bert_input.reshape(shape), bert_mask.reshape(shape), bert_segments.reshape(shape), self.maxqlen - 1, self._layers,
                                   mu, sigma, self.config["cls"], lin1.weight.detach().contiguous(), lin1.bias.detach(), w2, b2)
        return out.view(-1, 1)


class CEDRKNRM(Reranker):
    """MacAvaney, Yates, Cohan, Goharian. CEDR: Contextualized Embeddings for Document Ranking. SIGIR 2019 (reference CEDRKNRM.py:188-203).
    The first eight keys are the reference's options; microbatch / compute_dtype / skip_padding belong to this engine (as in ptBERTMaxP)."""

    module_name = "CEDRKNRM"
    config_spec = {"pretrained": "electra-base", "mus": [-0.9, -0.7, -0.5, -0.3, -0.1, 0.1, 0.3, 0.5, 0.7, 0.9], "sigma": 0.1,
                   "gradkernels": True, "hidden_dropout_prob": 0.1, "simmat_layers": list(range(13)), "combine_hidden": 1024, "cls": "avg",
                   "microbatch": 256, "compute_dtype": "fp16", "skip_padding": True}

    def build_model(self):
        if not hasattr(self, "model"):
            self.model = CEDRKNRM_Class(self.extractor, self.config)
        return self.model

    def score(self, d):
        return [self.model(d["pos_bert_input"], d["pos_mask"], d["pos_seg"]).view(-1),
                self.model(d["neg_bert_input"], d["neg_mask"], d["neg_seg"]).view(-1)]

    def test(self, d):
        return self.model(d["pos_bert_input"], d["pos_mask"], d["pos_seg"]).view(-1)
