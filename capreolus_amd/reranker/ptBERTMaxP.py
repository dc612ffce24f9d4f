"""BERT-MaxP behind the reference plugin surface (capreolus/reranker/ptBERTMaxP.py:99-135), scored by the
gfx950 kernels in capreolus_amd/csrc/bert*.{hip,cuh} through the C ABI.

The reference keeps a transformers `AutoModelForSequenceClassification` in ``self.bert``; here a bare
parameter container with the same attribute tree (and therefore the same state_dict names:
``bert.bert.embeddings.*``, ``bert.bert.encoder.layer.N.*``, ``bert.bert.pooler.dense.*``,
``bert.classifier.*``) holds the weights, so reference checkpoints load unchanged.  BERT-architecture checkpoints
(bert-base-uncased, Capreolus/bert-base-msmarco, ...) and RoBERTa-architecture ones (``bert.roberta.*``, ``bert.classifier.dense|out_proj.*``;
token types zeroed as ptBERTMaxP.py:57-58 does) are supported; the ELECTRA variants raise, as they do in the reference as written.
"""
import torch
from torch import nn

from .. import engine
from . import Reranker


class _Box(nn.Module):
    """A named bag of sub-modules (gives the HF attribute paths without any HF code)."""

    def __init__(box, **mods):  # noqa: N805  ("self" is one of the HF attribute names)
        super().__init__()
        for k, v in mods.items():
            setattr(box, k, v)


class _Embeddings(_Box):
    """transformers 4.9 (the reference's pin) keeps `position_ids` as a persistent buffer: it is in the reference's checkpoints and
    the reference's load_weights key check (reranker/__init__.py:44-50) requires it, so checkpoints written here carry it; newer
    transformers do not save it, so a state_dict without it still loads (strict or not)."""

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        if prefix + "position_ids" in missing_keys:
            missing_keys.remove(prefix + "position_ids")


def _encoder_layer(H, F):
    return _Box(
        attention=_Box(self=_Box(query=nn.Linear(H, H), key=nn.Linear(H, H), value=nn.Linear(H, H)),
                       output=_Box(dense=nn.Linear(H, H), LayerNorm=nn.LayerNorm(H, eps=1e-12))),
        intermediate=_Box(dense=nn.Linear(H, F)),
        output=_Box(dense=nn.Linear(F, H), LayerNorm=nn.LayerNorm(H, eps=1e-12)),
    )


def bert_body(hidden=768, layers=12, heads=12, ffn=3072, vocab=30522, max_pos=512, type_vocab=2, pooler=True):
    """Parameter tree of transformers.BertModel (state_dict names as HF); pooler=False: of transformers.ElectraModel, which is the same
    encoder without a pooler (for checkpoints whose embedding size equals the hidden size)."""
    mods = dict(
        embeddings=_Embeddings(word_embeddings=nn.Embedding(vocab, hidden, padding_idx=0), position_embeddings=nn.Embedding(max_pos, hidden),
                        token_type_embeddings=nn.Embedding(type_vocab, hidden), LayerNorm=nn.LayerNorm(hidden, eps=1e-12)),
        encoder=_Box(layer=nn.ModuleList([_encoder_layer(hidden, ffn) for _ in range(layers)])),
    )
    mods["embeddings"].register_buffer("position_ids", torch.arange(max_pos).expand((1, -1)).clone())
    if pooler:
        mods["pooler"] = _Box(dense=nn.Linear(hidden, hidden))
    body = _Box(**mods)
    body.num_attention_heads = heads
    return body


def bert_container(hidden=768, layers=12, heads=12, ffn=3072, vocab=30522, max_pos=512, type_vocab=2):
    """Parameter tree of BertForSequenceClassification(num_labels=2) (state_dict names as HF)."""
    body = bert_body(hidden, layers, heads, ffn, vocab, max_pos, type_vocab)
    box = _Box(bert=body, classifier=nn.Linear(hidden, 2))
    box.num_attention_heads = heads
    return box


def roberta_container(hidden=768, layers=12, heads=12, ffn=3072, vocab=50265, max_pos=514, type_vocab=1, pad_token_id=1, layer_norm_eps=1e-5):
    """Parameter tree of transformers.RobertaForSequenceClassification(num_labels=2) (state_dict names as HF): the BERT encoder under the
    attribute `roberta`, no pooler, a `dense -> tanh -> out_proj` head on the first token."""
    body = bert_body(hidden, layers, heads, ffn, vocab, max_pos, type_vocab, pooler=False)
    for ln in [body.embeddings.LayerNorm] + [m for lyr in body.encoder.layer for m in (lyr.attention.output.LayerNorm, lyr.output.LayerNorm)]:
        ln.eps = layer_norm_eps
    box = _Box(roberta=body, classifier=_Box(dense=nn.Linear(hidden, hidden), out_proj=nn.Linear(hidden, 2)))
    box.num_attention_heads, box.pad_token_id, box.layer_norm_eps = heads, pad_token_id, layer_norm_eps
    return box


class PTBERTMaxP_Class(nn.Module):
    def __init__(self, extractor, config):
        super().__init__()
        self.extractor = extractor
        self.config = config
        pre = config["pretrained"]
        self.is_roberta = False
        if isinstance(pre, dict):            # explicit geometry, weights loaded later with load_state_dict
            pre = dict(pre)
            self.is_roberta = pre.pop("arch", "bert") == "roberta"
            self.bert = roberta_container(**pre) if self.is_roberta else bert_container(**pre)
        elif isinstance(pre, str) and "electra" in pre:
            # (the reference's ELECTRA head, ptBERTMaxP.py:14-26, defines `call` instead of `forward`: it raises in the reference too)
            raise NotImplementedError(f"{pre}: only BERT- and RoBERTa-architecture sequence classifiers are scored by the MI355X engine")
        else:
            self.bert = self._from_hf(pre, config["hidden_dropout_prob"])
            self.is_roberta = hasattr(self.bert, "roberta")
        self._engine = None

    @staticmethod
    def _from_hf(name, hidden_dropout_prob):
        """Reads a local/cached HF checkpoint (no network here) and copies its tensors into the container."""
        from transformers import AutoModelForSequenceClassification

        if name == "bert-base-msmarco":
            name = "Capreolus/bert-base-msmarco"
        hf = AutoModelForSequenceClassification.from_pretrained(name, hidden_dropout_prob=hidden_dropout_prob)
        c = hf.config
        if c.model_type not in ("bert", "roberta") or c.hidden_act != "gelu" or c.num_labels != 2:
            raise NotImplementedError(f"{name}: unsupported architecture {c.model_type}/{c.hidden_act}/{c.num_labels} labels")
        if c.model_type == "roberta":
            box = roberta_container(c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.intermediate_size, c.vocab_size,
                                    c.max_position_embeddings, c.type_vocab_size, c.pad_token_id, c.layer_norm_eps)
        else:
            box = bert_container(c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.intermediate_size, c.vocab_size,
                                 c.max_position_embeddings, c.type_vocab_size)
        missing = box.load_state_dict(hf.state_dict(), strict=False)
        if missing.missing_keys:
            raise RuntimeError(f"checkpoint lacks {missing.missing_keys}")
        return box

    def _params(self):
        p = {k: v for k, v in self.bert.state_dict(keep_vars=True).items() if not k.endswith("position_ids")}
        if not self.is_roberta:
            return p
        # the engine addresses the encoder by the BERT names: same tensors, RoBERTa's head = pooler + classifier under other names
        q = {"bert." + k[len("roberta."):]: v for k, v in p.items() if k.startswith("roberta.")}
        q["bert.pooler.dense.weight"], q["bert.pooler.dense.bias"] = p["classifier.dense.weight"], p["classifier.dense.bias"]
        q["classifier.weight"], q["classifier.bias"] = p["classifier.out_proj.weight"], p["classifier.out_proj.bias"]
        return q

    def forward(self, doc_input, doc_mask, doc_seg):
        if self.training:
            raise NotImplementedError(
                "capreolus_amd scores with hand-written inference kernels; fine-tuning (ptBERTMaxP.py:60-61) is not part of "
                "this engine. Call under model.eval() as PytorchTrainer.predict does."
            )
        if self.is_roberta:
            doc_seg = torch.zeros_like(doc_mask)  # "since roberta does not have segment input" (ptBERTMaxP.py:57-58); with it the
            # passage mask of the sum / avg aggregations is all-False: sum scores 0, avg 0/0 = NaN - in the reference alike
        return self.predict_step(doc_input, doc_mask, doc_seg)

    def predict_step(self, doc_input, doc_mask, doc_seg):
        P, S = self.extractor.config["numpassages"], self.extractor.config["maxseqlen"]
        B = doc_input.shape[0]
        if self._engine is None:
            self._engine = engine.BertEngine(self._params(), self.bert.num_attention_heads,
                                             microbatch=int(self.config.get("microbatch", 256)),
                                             compute_dtype=self.config.get("compute_dtype", "fp16"),
                                             skip_padding=bool(self.config.get("skip_padding", True)),
                                             ln_eps=self.bert.layer_norm_eps if self.is_roberta else 0.0,
                                             pos_pad_id=self.bert.pad_token_id if self.is_roberta else -1)
        else:
            self._engine.params = self._params()
        shape = (B, P, S)
        return self._engine.forward(doc_input.reshape(shape), doc_mask.reshape(shape), doc_seg.reshape(shape), self.config["aggregation"])


class PTBERTMaxP(Reranker):
    """Dai & Callan, Deeper Text Understanding for IR with Contextual Neural Language Modeling, SIGIR'19
    (reference ptBERTMaxP.py:99-122)."""

    module_name = "ptBERTMaxP"
    # the first three are the reference's options (ptBERTMaxP.py:114-122); microbatch / compute_dtype belong to this engine:
    # compute_dtype "fp16" (default: the type the reference's amp=pred autocast uses, trainer/pytorch.py:323-326; measured
    # 7.5e-4 relative error on BERT-base logits, inside the 1e-3 parity bar) or "bf16" (BASELINE.json's wording; same
    # MFMA rate, wider range, 7e-3 error)
    # skip_padding (default True): passages are encoded in length buckets (multiples of 32 tokens up to maxseqlen) - bit-identical logits, the
    # padded rows are simply not computed (engine.BertEngine.forward)
    config_spec = {"pretrained": "bert-base-uncased", "aggregation": "max", "hidden_dropout_prob": 0.1, "microbatch": 256,
                   "compute_dtype": "fp16", "skip_padding": True}

    @property
    def batch_coupled(self):
        """aggregation = avg divides every document's passage sum by the BATCH-wide passage count (reference ptBERTMaxP.py:92): the
        scores depend on the batch composition, so `PytorchTrainer.predict` must not merge DataLoader batches for it."""
        return self.config["aggregation"] == "avg"

    def build_model(self):
        self.model = PTBERTMaxP_Class(self.extractor, self.config)
        return self.model

    def score(self, d):
        return [self.model(d["pos_bert_input"], d["pos_mask"], d["pos_seg"]).view(-1),
                self.model(d["neg_bert_input"], d["neg_mask"], d["neg_seg"]).view(-1)]

    def test(self, d):
        return self.model(d["pos_bert_input"], d["pos_mask"], d["pos_seg"]).view(-1)
