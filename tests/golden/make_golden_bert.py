"""BERT-MaxP golden vectors: the REFERENCE PTBERTMaxP_Class (imported from /root/reference) driving
transformers' BertForSequenceClassification, on seeded weights (oracle/bert_port.random_weights) and
seeded BertPassage-shaped inputs.  Build container only; see make_golden.py."""
import os
from types import SimpleNamespace

import numpy as np
import torch

from capreolus_amd import synthetic
from oracle import bert_port

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: model dims, inputs
    "mini": dict(hidden=128, layers=2, heads=2, ffn=512, vocab=1000, max_pos=128, B=5, P=3, S=64, seed=11),
    "mini_s128": dict(hidden=192, layers=1, heads=3, ffn=256, vocab=1200, max_pos=128, B=3, P=2, S=128, seed=12),
    "base": dict(hidden=768, layers=12, heads=12, ffn=3072, vocab=30522, max_pos=512, B=3, P=4, S=256, seed=13),
}


def gen_bert(MAXP):
    from transformers import AutoModelForSequenceClassification, BertConfig, BertForSequenceClassification

    for name, c in CASES.items():
        cfg = BertConfig(num_labels=2, hidden_size=c["hidden"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                         intermediate_size=c["ffn"], vocab_size=c["vocab"], max_position_embeddings=c["max_pos"])
        w = bert_port.random_weights(c["hidden"], c["layers"], c["heads"], c["ffn"], c["vocab"], c["max_pos"], seed=c["seed"])
        orig = AutoModelForSequenceClassification.from_pretrained
        AutoModelForSequenceClassification.from_pretrained = staticmethod(lambda *a, **k: BertForSequenceClassification(cfg))
        try:
            out = {}
            rs = np.random.RandomState(c["seed"])
            batch = synthetic.make_bert_passages(rs, c["B"], c["P"], c["S"], vocab=c["vocab"], same_query=True)
            ti = {k: torch.from_numpy(v) for k, v in batch.items()}
            for agg in ("max", "first", "sum", "avg"):
                model = MAXP.PTBERTMaxP_Class(
                    SimpleNamespace(config={"numpassages": c["P"], "maxseqlen": c["S"]}),
                    {"pretrained": "bert-base-uncased", "aggregation": agg, "hidden_dropout_prob": 0.1})
                missing = model.bert.load_state_dict(w, strict=False)
                assert not [k for k in missing.missing_keys if "position_ids" not in k], missing
                model.eval()
                with torch.no_grad():
                    out["ref_" + agg] = model(ti["pos_bert_input"], ti["pos_mask"], ti["pos_seg"]).view(-1).numpy().astype(np.float32)
                    if agg == "max":
                        logits = model.bert(ti["pos_bert_input"].reshape(-1, c["S"]), attention_mask=ti["pos_mask"].reshape(-1, c["S"]),
                                            token_type_ids=ti["pos_seg"].reshape(-1, c["S"]))[0]
                        out["ref_passage_logits"] = logits.numpy().astype(np.float32)
        finally:
            AutoModelForSequenceClassification.from_pretrained = orig
        np.savez_compressed(
            os.path.join(HERE, f"bert_{name}.npz"), weight_seed=np.int64(c["seed"]),
            dims=np.array([c["hidden"], c["layers"], c["heads"], c["ffn"], c["vocab"], c["max_pos"]], dtype=np.int64),
            pos_bert_input=batch["pos_bert_input"].astype(np.int32), pos_mask=batch["pos_mask"].astype(np.int8),
            pos_seg=batch["pos_seg"].astype(np.int8), **out)
        print("bert", name, {k: v[:3] for k, v in out.items() if k.startswith("ref_") and v.ndim == 1})
