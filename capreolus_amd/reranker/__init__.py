"""Reranker plugin surface, mirroring capreolus.reranker.Reranker (reference
capreolus/reranker/__init__.py:7-55): ``build_model``, ``score(d)``, ``test(d)``,
``save_weights`` / ``load_weights`` with the same checkpoint format (pickled state_dict minus
``embedding.weight`` / ``_nosave_`` keys + ``<fn>.optimizer``), so weights trained with the
reference load here and vice versa.

profane (the reference's module-graph library) is not a dependency: a reranker is constructed
directly from a config dict and an extractor-like object (anything with the attributes the model
reads: ``embeddings`` for KNRM/DRMM, ``config`` for BERT-MaxP).
"""
import os
import pickle


class Reranker:
    module_type = "reranker"
    module_name = None
    config_spec = {}  # key -> default (the reference's ConfigOption defaults)

    def __init__(self, config=None, extractor=None, trainer=None):
        cfg = dict(self.config_spec)
        unknown = set(config or {}) - set(cfg)
        if unknown:
            raise ValueError(f"unknown config options for {self.module_name}: {sorted(unknown)}")
        cfg.update(config or {})
        self.config = cfg
        self.extractor = extractor
        self.trainer = trainer

    # `supports_resident`: `test_resident` is meaningful (an interaction model over term-id rows; the BERT rerankers' inputs are
    # per (query, passage)).  `batch_coupled`: a pair's score depends on what else is in the batch (never true for the interaction
    # models; ptBERTMaxP with aggregation = avg, reference ptBERTMaxP.py:92) - `PytorchTrainer.predict` then keeps the DataLoader's
    # batches as they are.
    supports_resident = False
    batch_coupled = False
    # `supports_lists`: `test_lists(d, offsets)` / `test_resident_lists(store, pair_q, pair_d, offsets)` score whole candidate lists (pairs
    # laid out list after list, every list against its first pair's query) - KNRM and DRMM, whose kernels then gather every distinct
    # term of a LIST once instead of every distinct term of every document (csrc/lists.hip)
    supports_lists = False
    # `lists_bit_identical`: the list route's scores equal the per-pair route's bit for bit (what PytorchTrainer's `lists` = "exact" asks)
    lists_bit_identical = False
    # `lists_max_qlen`: query terms (the extractor's `maxqlen`) the list route takes - 8 for KNRM / DRMM / DRMM-TKS (two blocks of four
    # terms, csrc/lists.h), 4 for PACRR's MFMA list kernel; longer queries are scored by the per-pair kernels (any length)
    lists_max_qlen = 4

    def build_model(self):
        raise NotImplementedError

    def score(self, d):
        raise NotImplementedError

    def test(self, d):
        raise NotImplementedError

    def _score_pos_neg(self, d):
        """`score(d)` of the interaction models (reference e.g. KNRM.py:87-94): the positive and the negative document of every training
        pair through ONE model call - documents are independent rows, so the scores are those of two calls, at half the launches of a
        training step (a batch-32 step is launch-bound: PytorchTrainer replays it as one HIP graph, this halves its nodes)."""
        import torch

        q, idf, pos = d["query"], d["query_idf"], d["posdoc"]
        both = self.model(torch.cat([pos, d["negdoc"]]), torch.cat([q, q]), torch.cat([idf, idf])).view(-1)
        return [both[: pos.shape[0]], both[pos.shape[0]:]]

    def test_resident(self, store, pair_q, pair_d):
        """Scores (query row, document row) pairs of a device-resident `capreolus_amd.feeder.CandidateStore`.
        Default for the interaction models: the id rows are gathered ON THE DEVICE from the store's int32 tables (no host
        round trip, no DataLoader) and handed to `test`; KNRM and DRMM override this with kernels that read the tables directly."""
        import torch

        pq, pd = pair_q.long(), pair_d.long()
        return self.test({"query": store.q_table.index_select(0, pq).long(), "posdoc": store.d_table.index_select(0, pd).long(),
                          "query_idf": store.idf_table.index_select(0, pq).to(torch.float32)})

    def add_summary(self, summary_writer, niter):
        for name, weight in self.model.named_parameters():
            summary_writer.add_histogram(name, weight.data.cpu(), niter)

    @staticmethod
    def _saved_keys(state_dict):
        return {k: v for k, v in state_dict.items() if "embedding.weight" not in k and "_nosave_" not in k}

    @staticmethod
    def _portable_optimizer_state(optimizer):
        """The optimizer's state_dict as the reference's plain Adam would have written it: a graphed training step runs
        Adam(capturable=True, lr=<device scalar>) (PytorchTrainer `graph`), whose state_dict would otherwise carry a CUDA `lr` tensor,
        `capturable = True` and device-side step counters - a checkpoint the reference (or an eager run here) could not resume from."""
        import torch

        sd = optimizer.state_dict()
        groups = []
        for g in sd["param_groups"]:
            g = dict(g)
            if torch.is_tensor(g.get("lr")):
                g["lr"] = float(g["lr"])
            if g.get("capturable"):
                g["capturable"] = False
            groups.append(g)
        state = {k: {n: (v.detach().cpu() if n == "step" and torch.is_tensor(v) else v) for n, v in st.items()} for k, st in sd["state"].items()}
        return {"state": state, "param_groups": groups}

    def save_weights(self, weights_fn, optimizer):
        weights_fn = os.fspath(weights_fn)
        os.makedirs(os.path.dirname(weights_fn) or ".", exist_ok=True)
        with open(weights_fn, "wb") as outf:
            pickle.dump(self._saved_keys(self.model.state_dict()), outf, protocol=-1)
        with open(weights_fn + ".optimizer", "wb") as outf:
            pickle.dump(self._portable_optimizer_state(optimizer), outf, protocol=-1)

    def load_weights(self, weights_fn, optimizer):
        import torch

        weights_fn = os.fspath(weights_fn)
        with open(weights_fn, "rb") as f:
            d = pickle.load(f)
        # (position_ids: a persistent buffer only in the transformers the reference pins; newer checkpoints do not carry it)
        missing = {k for k in set(self._saved_keys(self.model.state_dict())) - set(d) if not k.endswith("position_ids")}
        if missing:
            raise RuntimeError("loading state_dict with keys that do not match current model: %s" % missing)
        self.model.load_state_dict(d, strict=False)
        # what kind of Adam the caller runs (a captured training step needs capturable = True and a device scalar as the learning rate)
        # survives the load: the checkpoint is the plain kind whoever wrote it
        kinds = [(bool(g.get("capturable")), g["lr"].device if torch.is_tensor(g.get("lr")) else None) for g in optimizer.param_groups]
        with open(weights_fn + ".optimizer", "rb") as f:
            optimizer.load_state_dict(pickle.load(f))
        for g, (capturable, lr_dev) in zip(optimizer.param_groups, kinds):
            if lr_dev is not None and not torch.is_tensor(g["lr"]):
                g["lr"] = torch.tensor(float(g["lr"]), device=lr_dev)
            if capturable:
                g["capturable"] = True
                for p in g["params"]:
                    st = optimizer.state.get(p)
                    if st and "step" in st:
                        st["step"] = torch.as_tensor(st["step"], dtype=torch.float32).to(p.device)


from .KNRM import KNRM, KNRM_class  # noqa: E402,F401
from .DRMM import DRMM, DRMM_class  # noqa: E402,F401

from .DRMMTKS import DRMMTKS, DRMMTKS_class  # noqa: E402,F401
from .ConvKNRM import ConvKNRM, ConvKNRM_class  # noqa: E402,F401
from .PACRR import PACRR, PACRR_class  # noqa: E402,F401
from .ptBERTMaxP import PTBERTMaxP, PTBERTMaxP_Class  # noqa: E402,F401
from .CEDRKNRM import CEDRKNRM, CEDRKNRM_Class  # noqa: E402,F401

registry = {"KNRM": KNRM, "DRMM": DRMM, "DRMMTKS": DRMMTKS, "PACRR": PACRR, "ConvKNRM": ConvKNRM, "ptBERTMaxP": PTBERTMaxP, "CEDRKNRM": CEDRKNRM}
