"""Prediction loop behind the reference trainer surface (capreolus/trainer/pytorch.py:310-377),
with the one thing the reference lacks: sharding of the candidate lists over the GPUs of a node.

Single process: identical control flow to the reference `PytorchTrainer.predict` — DataLoader over
the prediction sampler, last incomplete batch filled by repetition (:355-377), ``reranker.test``
per batch, scores rounded through float16 (:346-348), TREC run written with the reference's
ordering (searcher/__init__.py:48-58).

One process per GPU (``torch.distributed`` initialised, backend "nccl" = RCCL): every rank scores
a contiguous block of *queries* (a query's candidates stay on one GPU) and the fp32 score vectors
are exchanged with a single all_gather at the end.  No other collective is on the
path.
"""
import copy
import itertools
import math
import os

import numpy as np
import torch

from ..run_io import write_trec_run


def _preds_dict(keys, scores):
    """{qid: {docid: score}} with the scores rounded to float16 - what the reference hands to pytrec_eval and to the run file
    (trainer/pytorch.py:346-348).  One vectorised conversion: `score.astype(np.float16).item()` per pair was most of predict_resident's time."""
    vals = np.asarray(scores).astype(np.float16).tolist()
    preds = {}
    for (qid, docid), v in zip(keys, vals):
        preds.setdefault(qid, {})[docid] = v
    return preds


def shard_bounds(sizes, world):
    """Contiguous split of items with the given sizes into `world` blocks of near-equal total size.
    Returns world+1 boundaries (indices into the item list)."""
    total = sum(sizes)
    bounds, acc, nxt = [0], 0, 1
    for i, s in enumerate(sizes):
        # close block nxt-1 before item i if that gets closer to its ideal end
        while nxt < world and acc + s / 2.0 > total * nxt / world:
            bounds.append(i)
            nxt += 1
        acc += s
    while len(bounds) < world:
        bounds.append(len(sizes))
    bounds.append(len(sizes))
    return bounds


def shard_pred_data(pred_data, rank, world):
    """The part of a prediction sampler this rank scores, plus (offset, count, total) in samples.

    Samplers that expose ``qid_to_docids`` (the reference PredSampler, sampler/__init__.py:207-264)
    are split by query; anything else is split into contiguous sample ranges.
    """
    if world == 1:
        n = len(pred_data)
        return pred_data, 0, n, n
    q2d = getattr(pred_data, "qid_to_docids", None)
    if q2d is not None:
        qids = list(q2d.keys())
        sizes = [len(q2d[q]) for q in qids]
        b = shard_bounds(sizes, world)
        mine = qids[b[rank]:b[rank + 1]]
        part = copy.copy(pred_data)
        part.qid_to_docids = {q: q2d[q] for q in mine}
        offset = sum(sizes[: b[rank]])
        return part, offset, sum(sizes[b[rank]:b[rank + 1]]), sum(sizes)
    n = len(pred_data)
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world

    class _Slice(torch.utils.data.IterableDataset):
        def __iter__(self_inner):
            return itertools.islice(iter(pred_data), lo, hi)

        def __len__(self_inner):
            return hi - lo

    return _Slice(), lo, hi - lo, n


class PytorchTrainer:
    module_name = "pytorch"
    config_spec = {  # reference defaults, trainer/pytorch.py:24-44
        "batch": 32, "evalbatch": 0, "niters": 20, "itersize": 512, "gradacc": 1, "lr": 0.001, "softmaxloss": False,
        "fastforward": False, "validatefreq": 1, "multithread": False, "boardname": "default", "warmupiters": 0,
        "decay": 0.0, "decayiters": 3, "decaytype": None, "amp": None, "seed": 123,
        # this engine's own: `predict` hands the scorer one batch per `coalesce` pairs instead of one per DataLoader batch (the
        # reference default evalbatch = batch = 32 would be 32-workgroup launches on a 256-CU chip); 0 = one call per DataLoader
        # batch with the reference's fill-by-repetition of the last one.  Pairs are scored independently, so the scores do not depend
        # on it - except for rerankers that say `batch_coupled` (ptBERTMaxP aggregation = avg divides by a batch-wide count,
        # ptBERTMaxP.py:92): those are never coalesced.  A merged batch is also capped at `coalesce_bytes` of input tensors.
        "coalesce": 16384, "coalesce_bytes": 256 << 20,
        # `resident` (default on): a sampler with the PredSampler contract scored by a reranker that can read a device-resident
        # candidate store is tokenised ONCE - the first `predict` of a sampler walks it like the DataLoader would and uploads the
        # distinct query / document id rows as int32 tables (capreolus_amd.feeder.CandidateStore); that call and every later one on the
        # same sampler (the dev set after every training iteration, RerankTask's repeated predict) score it by index pairs without
        # touching the host per sample (SURVEY.md row N1 through the reference's own call site, trainer/pytorch.py:310-353).
        # `resident_verify`: how a later call recognises that the sampler still holds what the store was built from.  "auto" (default):
        # up to 100,000 pairs every docid of every list is compared with the plan's own copy (0.2 ms per 64,000 - an in-place edit of a
        # candidate list cannot go unnoticed), above that "sampled" (the list objects' identities, sizes and eight docids each); "full"
        # hashes every docid at any size.
        "resident": True, "resident_verify": "auto",
        # `lists`: which rerankers the resident route scores as whole candidate lists (csrc/lists.hip: every distinct term of a LIST
        # gathered once).  "always" (default): every reranker that takes lists - KNRM, DRMM, DRMM-TKS, PACRR; this is the route
        # `bench.py` times.  DRMM / DRMM-TKS / PACRR list scores equal their per-pair scores bit for bit; KNRM's pooling sums run in
        # another order (1e-6 relative) - BOTH of its routes are pinned on the reference's fp16 predictions and run order over a
        # multi-query run (tests/golden/knrm_multiquery.npz).  "exact": only the bit-identical ones (KNRM pair by pair); "never": per-pair
        # kernels only.
        "lists": "always",
        # `graph` (default on): a training step - score() on positives and negatives, loss, backward, Adam - is captured ONCE as a HIP
        # graph and replayed per batch (at batch 32 a step is ~40 launches of microsecond kernels, i.e. host time).
        # Needs a GPU, gradacc = 1 and no loss scaling (amp = train / both); anything else, and batches of another shape, run eagerly.
        "graph": True,
        # `fused` (default on): rerankers that bring a whole training step as device kernels (`fused_train_step`: KNRM with a single-Linear
        # `combine` - score(pos), score(neg), the pairwise loss, backward and Adam's update in two launches, capamd_knrm_train_step) train
        # through it with the PLAIN torch.optim.Adam of the reference as the optimizer object (its state_dict / checkpoints unchanged,
        # bias corrections in double on the host).  Same conditions as `graph`; a reranker or configuration without one falls back to it.
        "fused": True,
    }
    # amp = "pred" / "both" at prediction time (reference :323-326, 343: autocast around `reranker.test`) selects nothing here: the
    # interaction kernels (KNRM, DRMM, ...) compute in fp32 and the BERT encoder already runs on 16-bit operands - the scores are
    # those of the non-autocast reference within the parity bar either way.

    LISTS_PAIR_BYTES = 1 << 30      # per-pair workspace bytes one whole-list scoring call may ask for (see _score_store)

    def __init__(self, config=None):
        cfg = dict(self.config_spec)
        unknown = set(config or {}) - set(cfg)
        if unknown:
            raise ValueError(f"unknown config options for trainer: {sorted(unknown)}")
        cfg.update(config or {})
        self.config = cfg
        self.build()

    def build(self):
        """The reference's sanity checks and seeding (trainer/pytorch.py:47-74)."""
        c = self.config
        if c["batch"] < 1:
            raise ValueError("batch must be >= 1")
        if c["evalbatch"] < 0:
            raise ValueError("evalbatch must be 0 (to use the training batch size) or  >= 1")
        if c["niters"] <= 0:
            raise ValueError("niters must be > 0")
        if c["niters"] < c["validatefreq"]:
            raise ValueError("niters must be equal or greater than validatefreq")
        if c["itersize"] < c["batch"]:
            raise ValueError("itersize must be >= batch")
        if c["gradacc"] < 1 or not float(c["gradacc"]).is_integer():
            raise ValueError("gradacc must be an integer >= 1")
        if c["lr"] <= 0:
            raise ValueError("lr must be > 0")
        if c["amp"] not in (None, "train", "pred", "both"):
            raise ValueError("amp must be one of: None, train, pred, both")
        if c["decaytype"] not in (None, "exponential", "linear"):
            raise ValueError("decaytype must be one of: None, exponential, linear")
        if c["resident_verify"] not in ("auto", "sampled", "full"):
            raise ValueError("resident_verify must be one of: auto, sampled, full")
        if c["lists"] not in ("exact", "always", "never"):
            raise ValueError("lists must be one of: exact, always, never")
        torch.manual_seed(c["seed"])
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(c["seed"])

    @property
    def n_batch_per_iter(self):
        return (self.config["itersize"] // self.config["batch"]) or 1     # reference trainer/__init__.py:74-76

    def lr_multiplier(self, step):
        """Learning-rate factor of optimisation step `step`: linear ramp over the warm-up steps, then the configured decay measured
        in iterations since the end of the warm-up (same schedule as the reference's LambdaLR, trainer/__init__.py:98-109)."""
        c, per_iter = self.config, self.n_batch_per_iter
        ramp = c["warmupiters"] * per_iter
        if ramp > 0 and step <= ramp:
            return min(1, (step + 1) / ramp)
        iters_after = (step - ramp) / per_iter
        decay = {"exponential": lambda: c["decay"] ** (iters_after / c["decayiters"]),
                 "linear": lambda: 1 / (1 + c["decay"] * iters_after)}.get(c["decaytype"])
        return decay() if decay else 1

    def _set_lr(self, step):
        # what `LambdaLR.step(epoch=step)` leaves behind in the reference (:118-120): lr = base lr x multiplier(step)
        for group in self.optimizer.param_groups:
            lr = self.config["lr"] * self.lr_multiplier(step)
            if torch.is_tensor(group["lr"]):
                group["lr"].fill_(lr)        # (capturable Adam: the captured step reads the learning rate from this device scalar)
            else:
                group["lr"] = lr

    # ---- training (reference trainer/pytorch.py:76-122, 189-300) -------------------------
    @staticmethod
    def pair_hinge_loss(pos_neg_scores):
        """reference reranker/common.py:101-103: MarginRankingLoss(margin=1, reduction="mean") with target +1."""
        pos, neg = pos_neg_scores
        return torch.clamp(1.0 - (pos - neg), min=0).mean()

    @staticmethod
    def pair_softmax_loss(pos_neg_scores):
        """reference reranker/common.py:96-98."""
        scores = torch.stack(pos_neg_scores, dim=1)
        return torch.mean(1.0 - scores.softmax(dim=1)[:, 0])

    # ---- one training step as one HIP graph -------------------------------------------------------------------------------------
    def _graph_allowed(self):
        return bool(self.config["graph"]) and self.device.type == "cuda" and self.config["gradacc"] == 1 and self.scaler is None and \
            not getattr(self, "_graph_failed", False)

    def _capture_train_step(self, reranker, tens, other, sig):
        """Captures score -> loss -> backward -> optimizer step on static copies of one batch's tensors.  The eager warm-up the capture
        needs (lazy optimizer state, library handles, autotuned convolutions) runs on a side stream and is then UNDONE - parameters and
        Adam moments are put back in place - so that training takes exactly the steps it would take eagerly."""
        from .. import engine

        static = {k: v.to(self.device).clone() for k, v in tens.items()}
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        snap_p = [p.detach().clone() for p in params]
        snap_s = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.optimizer.state.get(p, {}).items()} for p in params}

        def step():
            loss = self.loss(reranker.score({**other, **static}))
            loss.backward()
            self.optimizer.step()
            return loss

        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=self.device)
        rng = torch.cuda.get_rng_state(self.device)
        try:
            side.wait_stream(cur)
            with torch.cuda.stream(side), engine.deferred_status(self.device):
                for _ in range(2):
                    self.optimizer.zero_grad(set_to_none=True)
                    step()
            cur.wait_stream(side)
        finally:
            # whatever the warm-up did - or failed half-way through (the caller then trains eagerly) - training continues from where it
            # stood: parameters, Adam state and the device RNG (dropout) are put back
            torch.cuda.synchronize(self.device)
            with torch.no_grad():
                for p, sp in zip(params, snap_p):
                    p.copy_(sp)
                for p in params:
                    for k, v in self.optimizer.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            old = snap_s[p].get(k)
                            v.copy_(old) if old is not None else v.zero_()
            torch.cuda.set_rng_state(rng, self.device)
            self.optimizer.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with engine.deferred_status(self.device), torch.cuda.graph(graph):
            loss = step()
        return {"sig": sig, "reranker": reranker, "static": static, "graph": graph, "loss": loss}

    def _fused_allowed(self, reranker):
        """The reranker brings its training step as device kernels AND takes this configuration with them (otherwise the captured-graph
        route, with its own kind of Adam, is the better second choice than a plain Adam stepping eagerly)."""
        available = getattr(reranker, "fused_step_available", None)
        return bool(self.config["fused"]) and self.device.type == "cuda" and self.config["gradacc"] == 1 and self.scaler is None and \
            callable(getattr(reranker, "fused_train_step", None)) and callable(available) and bool(available(self.config["batch"])) and \
            not getattr(self, "_fused_failed", False)

    def _fused_step(self, reranker, batch):
        """The batch's step as the reranker's own device kernels; None when its configuration has none (then: graph / eager)."""
        try:
            loss = reranker.fused_train_step(batch, self.optimizer, softmax=bool(self.config["softmaxloss"]))
        except NotImplementedError:
            loss = None
        if loss is None:
            self._fused_failed = True
        return loss

    def _graphed_step(self, reranker, batch):
        """Replays the captured step on `batch`; None when this batch cannot take the graph (another shape: the short last batch)."""
        from .. import engine

        tens = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        other = {k: v for k, v in batch.items() if not torch.is_tensor(v)}
        sig = tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(tens.items()))
        gs = getattr(self, "_train_graph", None)
        if gs is not None and (gs["reranker"] is not reranker or gs["optimizer"] is not self.optimizer):
            gs = self._train_graph = None
        if gs is None:
            try:
                gs = self._capture_train_step(reranker, tens, other, sig)
            except RuntimeError:
                self._graph_failed = True        # something in this model's step cannot be captured: eager from here on
                torch.cuda.synchronize(self.device)
                return None
            gs["optimizer"] = self.optimizer
            self._train_graph = gs
        elif gs["sig"] != sig:
            return None
        for k, v in tens.items():
            gs["static"][k].copy_(v, non_blocking=True)
        gs["graph"].replay()       # (the kernels' status word is read once per iteration, by single_train_iteration)
        return gs["loss"].detach().clone()

    def single_train_iteration(self, reranker, train_dataloader, cur_iter=1):
        """`itersize // batch` batches with gradient accumulation, the per-step learning-rate schedule and (amp = train / both,
        on a GPU) autocast + loss scaling around the small trainable layers (reference :76-122).  The interaction kernels
        compute in fp32 whatever `amp` says."""
        n_batch_per_iter = self.n_batch_per_iter
        cur_step = cur_iter * n_batch_per_iter
        graphed = self._graph_allowed() or getattr(self, "_use_fused", False)
        if graphed:      # no device -> host read inside the iteration: data-dependent errors of the kernels are raised once, after it
            from .. import engine

            with engine.deferred_status(self.device):
                return self._train_batches(reranker, train_dataloader, n_batch_per_iter, cur_step, True)
        return self._train_batches(reranker, train_dataloader, n_batch_per_iter, cur_step, False)

    def _train_batches(self, reranker, train_dataloader, n_batch_per_iter, cur_step, graphed):
        losses, since_update, replayed = [], 0, False
        fused = getattr(self, "_use_fused", False)
        for bi, batch in enumerate(train_dataloader):
            batch = {k: v.to(self.device) if torch.is_tensor(v) else v for k, v in batch.items()}
            done = None
            if fused:
                done = self._fused_step(reranker, batch)
                if done is None:
                    # the reranker's device step does not cover this configuration (engine.AdamStep: an optimizer parameter it does not
                    # update, a batch it cannot take): the captured-graph route is the second choice - from this batch on, not eager
                    # steps for the rest of training
                    fused = self._use_fused = False
                    graphed = self._graph_allowed()
            if done is None and graphed and not fused:
                done = self._graphed_step(reranker, batch)
                replayed = replayed or done is not None
            if done is not None:
                losses.append(done)
            else:
                if graphed and not fused and since_update == 0:
                    # a batch the graph cannot take (another shape): after a replay every p.grad IS the graph's static gradient tensor,
                    # still holding the previous step's values - backward() would add to them
                    self.optimizer.zero_grad(set_to_none=True)
                with self._train_autocast():
                    loss = self.loss(reranker.score(batch))
                losses.append(loss.detach())
                (self.scaler.scale(loss) if self.scaler else loss).backward()
                since_update += 1
                if since_update == self.config["gradacc"]:
                    since_update = 0
                    if self.scaler:
                        self.scaler.step(self.optimizer)
                        self.scaler.update()
                    else:
                        self.optimizer.step()
                    self.optimizer.zero_grad()
            if (bi + 1) % n_batch_per_iter == 0:
                break
            self._set_lr(cur_step)
            cur_step += 1
        if replayed:
            self._mark_parameters_changed()
        return torch.stack(losses).mean()

    def _mark_parameters_changed(self):
        """A graph replay updates the parameters behind autograd's back: `tensor._version` - what the engine's weight-derived caches
        (PackedEmbedding, ConvKNRM's projection tables, the BERT blob) are keyed on - does not move.  An exact in-place no-op (x * 1)
        on every trained parameter bumps it, so that the next predict() rebuilds what was derived from the old weights."""
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        with torch.no_grad():
            torch._foreach_mul_(params, 1.0)

    @staticmethod
    def _early_stopping_paths(train_output_path, dev_output_path):
        """reference trainer/__init__.py:78-90"""
        weights, info = os.path.join(train_output_path, "weights"), os.path.join(train_output_path, "info")
        for p in (dev_output_path, weights, info):
            os.makedirs(p, exist_ok=True)
        return os.path.join(train_output_path, "dev.best"), weights, os.path.join(info, "loss.txt"), os.path.join(dev_output_path, "metrics.json")

    @staticmethod
    def load_loss_file(fn):
        """The per-iteration mean losses of `info/loss.txt` (one `<iteration> <loss>` record per line, iteration numbers 0, 1, 2, ...
        - the layout the reference writes, trainer/__init__.py:22-48).  A gap or repeat in the numbering means two runs wrote to the
        same directory: IOError."""
        with open(fn, "rt") as f:
            records = [ln.split() for ln in f if ln.strip()]
        if [int(r[0]) for r in records] != list(range(len(records))):
            raise IOError(f"{fn}: iteration numbers are not 0..{len(records) - 1} in order (two writers?)")
        return [float(r[1]) for r in records]

    def fastforward_training(self, reranker, weights_path, loss_fn, metric_fn):
        """Resume from the last iteration whose weights were saved (reference :76-122 of trainer/pytorch.py `fastforward_training`):
        returns (next iteration, best metrics so far); (0, {}) when nothing usable is on disk."""
        import json

        if not (os.path.exists(weights_path) and os.path.exists(loss_fn)):
            return 0, {}
        try:
            loss = self.load_loss_file(loss_fn)
            with open(metric_fn, "rt") as f:
                metrics = json.load(f)
        except (IOError, ValueError):
            return 0, {}
        last = len(loss) - 1
        try:
            reranker.load_weights(os.path.join(weights_path, f"{last}.p"), self.optimizer)
            return last + 1, metrics
        except Exception:  # noqa: BLE001  (as the reference: any failure means "start over")
            return 0, {}

    def train(self, reranker, train_dataset, train_output_path, dev_data, dev_output_path, qrels, metric="ndcg_cut_20", relevance_level=1):
        """Pairwise training with validation on `dev_data` every `validatefreq` iterations, `dev.best` checkpointing and
        `fastforward` resume (reference :189-300).  Metrics: trec_eval-style ndcg_cut_k from capreolus_amd.run_io (`metric` =
        "ndcg_cut_<k>"); `relevance_level` does not change it - pytrec_eval's ndcg_cut uses the graded judgments whatever the
        level is (evaluator.py:55-85: the level applies to the binary metrics).  Returns the list of per-iteration mean losses."""
        import contextlib
        import json

        from ..run_io import mean_ndcg_cut

        if not metric.startswith("ndcg_cut_"):
            raise ValueError("this trainer validates with ndcg_cut_<k>")
        k = int(metric.rsplit("_", 1)[1])
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        model = reranker.model.to(self.device)
        if self.config["amp"] in ("both", "train") and self.device.type == "cuda":
            self._train_autocast = lambda: torch.autocast("cuda", dtype=torch.float16)
            self.scaler = torch.amp.GradScaler("cuda")
        else:
            self._train_autocast, self.scaler = contextlib.nullcontext, None
        self._train_graph, self._graph_failed, self._fused_failed = None, False, False
        self._use_fused = self._fused_allowed(reranker)
        if self._use_fused:            # the reranker's own step kernels update parameters and moments of the reference's plain Adam
            self.optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=self.config["lr"])
        elif self._graph_allowed():    # the captured step needs Adam's device-side step count and a device scalar as the learning rate
            self.optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()),
                                              lr=torch.tensor(float(self.config["lr"]), device=self.device), capturable=True)
        else:
            self.optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=self.config["lr"])
        self._set_lr(0)                      # LambdaLR's construction applies multiplier(0)
        self.loss = self.pair_softmax_loss if self.config["softmaxloss"] else self.pair_hinge_loss
        loader = torch.utils.data.DataLoader(train_dataset, batch_size=self.config["batch"], pin_memory=False,   # (see predict)
                                             num_workers=1 if self.config["multithread"] else 0)
        train_output_path, dev_output_path = os.fspath(train_output_path), os.fspath(dev_output_path)
        best_fn, weights_path, loss_fn, metric_fn = self._early_stopping_paths(train_output_path, dev_output_path)
        initial_iter, metrics = self.fastforward_training(reranker, weights_path, loss_fn, metric_fn) if self.config["fastforward"] else (0, {})
        best, train_loss = metrics.get(metric, -np.inf), []
        if initial_iter > 0:
            train_loss = self.load_loss_file(loss_fn)
            if initial_iter < self.config["niters"]:       # skip the batches the finished iterations consumed (reference :258-264)
                for i, _ in enumerate(loader):
                    if i + 1 == initial_iter * self.n_batch_per_iter:
                        break
        for niter in range(initial_iter + 1, self.config["niters"] + 1):
            model.train()
            train_loss.append(float(self.single_train_iteration(reranker, loader, cur_iter=niter)))
            if self.config["fastforward"]:
                reranker.save_weights(os.path.join(weights_path, f"{niter}.p"), self.optimizer)
            if niter % self.config["validatefreq"] == 0:
                preds = self.predict(reranker, dev_data, os.path.join(dev_output_path, f"{niter}.run"))
                score = mean_ndcg_cut(qrels, preds, k)
                if score > best:
                    best = score
                    reranker.save_weights(best_fn, self.optimizer)
                    with open(metric_fn, "wt") as f:
                        json.dump({metric: score}, f)
            with open(loss_fn, "wt") as f:
                f.write("\n".join(f"{i} {l}" for i, l in enumerate(train_loss)))
        return train_loss

    def load_best_model(self, reranker, train_output_path):
        """reference :302-308."""
        self.optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, reranker.model.parameters()), lr=self.config["lr"])
        reranker.load_weights(os.path.join(os.fspath(train_output_path), "dev.best"), self.optimizer)

    def fill_incomplete_batch(self, batch, batch_size=None):
        """Repeat-pad a short final batch (reference trainer/pytorch.py:355-377)."""
        batch_size = batch_size or self.config["batch"]
        n = len(batch["qid"])
        reps, diff = math.ceil(batch_size / n), batch_size - n

        def pad(v):
            if isinstance(v, np.ndarray) or torch.is_tensor(v):
                v = v.repeat((reps,) + (1,) * (len(v.shape) - 1))
            else:
                v = v + [v[0]] * diff
            return v[:batch_size]

        return {k: pad(v) for k, v in batch.items()}

    # ---- the resident route of `predict` ---------------------------------------------------------------------------------------
    def _sampler_fingerprint(self, pred_data):
        """What identifies a prediction sampler's content: per query its id and candidate count plus - `resident_verify` = "sampled", the
        default - the docid list object and eight evenly spaced docids of it, or - "full" - every docid (64,000 docids hash in ~3 ms, more
        than the scoring takes).  None for samplers without `qid_to_docids`, which take the DataLoader route.  A list edited in place
        between the sampled positions is the one change "sampled" cannot see: call `forget_candidate_stores()` (or use "full") then."""
        q2d = getattr(pred_data, "qid_to_docids", None)
        if not isinstance(q2d, dict) or not q2d:
            return None
        try:
            if self.config["resident_verify"] == "full":      # ("auto" adds its element-wise comparison in _plan_still_mirrors)
                return (len(q2d), hash(tuple((q, tuple(d)) for q, d in q2d.items())))
            return (len(q2d), hash(tuple((q, len(d), id(d)) + tuple(d[:: max(1, (len(d) - 1) // 7)]) + (d[-1],) if len(d) else (q, 0) for q, d in q2d.items())))
        except TypeError:
            return None

    def _plan_still_mirrors(self, pred_data, plan):
        """resident_verify = "auto": the element-wise check of a plan whose lists were, when it was built, exactly the sampler's
        `qid_to_docids` lists (the PredSampler contract) and hold at most 100,000 pairs - tuple(list) == the plan's tuple, an identity
        comparison per docid.  Other plans rest on the sampled fingerprint alone."""
        extra = plan[4]
        if self.config["resident_verify"] != "auto" or not extra.get("mirrors_q2d") or int(extra["offsets"][-1]) > 100000:
            return True
        q2d = pred_data.qid_to_docids
        try:
            return all(tuple(q2d[q]) == ds for q, ds, _ in plan[3])
        except (KeyError, TypeError):
            return False

    def forget_candidate_stores(self):
        """Drops the device-resident candidate stores `predict` built (the next call on a sampler tokenises and uploads it again)."""
        self.__dict__.pop("_resident_plans", None)
        self.__dict__.pop("_eval_plan", None)

    def _resident_plan(self, pred_data, part, rank, world):
        """(store, pair_q, pair_d, groups) for this rank's part of `pred_data`, built on the first call for a sampler and kept for
        the next ones; None when the samples are not interaction-model id rows or a docid's row depends on the query it comes with."""
        from ..feeder import CandidateStore

        fp = self._sampler_fingerprint(pred_data)
        if fp is None:
            return None
        plans = self.__dict__.setdefault("_resident_plans", {})
        key = (id(pred_data), rank, world, str(self.device))
        hit = plans.get(key)
        if hit is not None and hit[0] == fp and hit[1]() is pred_data and self._plan_still_mirrors(pred_data, hit[2]):
            return hit[2]
        store, pq, pd, groups = CandidateStore(self.device), [], [], []
        # the same walk the DataLoader would do - once.  Per sample: two dictionary lookups and a comparison of the query row's BYTES with
        # the ones its qid came with first (np.array_equal on a 4-element row cost more than everything else in this loop); a document's
        # 800-element row is compared only when its docid comes a second time
        qrow_of, drow_of, rows_d = store.qrow, store.drow, store._d
        qbytes, cur_qid, cur_qrow, cur_docs = {}, object(), -1, None
        ndarray, last_q, last_i, last_sig = np.ndarray, None, None, None
        for sample in part:
            try:
                qid, docid, query, doc = sample["qid"], sample["posdocid"], sample["query"], sample["posdoc"]
            except KeyError:
                return None
            idf = sample.get("query_idf")
            if query is last_q and idf is last_i:        # (a sampler that hands out the same row objects per query: nothing to compare)
                sig = last_sig
            else:
                sig = (query.tobytes() if isinstance(query, ndarray) else np.asarray(query).tobytes(),
                       None if idf is None else idf.tobytes() if isinstance(idf, ndarray) and idf.dtype == np.float32 else np.asarray(idf, np.float32).tobytes())
                last_q, last_i, last_sig = query, idf, sig
            if qid != cur_qid:                # a new run of this qid's samples = a new list
                qrow = qrow_of.get(qid)
                if qrow is None:
                    qrow = store.add_query(qid, query, idf)
                    qbytes[qrow] = sig
                cur_qid, cur_qrow, cur_docs = qid, qrow, []
                groups.append([qid, cur_docs, len(pq)])
            if sig != qbytes[cur_qrow] and not (np.array_equal(store._q[cur_qrow], np.asarray(query)) and
                                                (idf is None or np.array_equal(store._idf[cur_qrow], np.asarray(idf, np.float32)))):
                return None       # a qid whose query row changes from sample to sample: not a candidate-store sampler
            drow = drow_of.get(docid)
            if drow is None:
                drow = drow_of[docid] = len(rows_d)
                rows_d.append(doc if isinstance(doc, ndarray) else np.asarray(doc))
            elif not np.array_equal(rows_d[drow], np.asarray(doc)):
                return None       # this extractor's document row depends on the query: not a candidate-store sampler
            cur_docs.append(docid)
            pq.append(cur_qrow)
            pd.append(drow)
        if not pq:
            return None
        store.finalize()
        import weakref

        counts = [len(ds) for _, ds, _ in groups]
        plan = (store, torch.as_tensor(np.asarray(pq, dtype=np.int32)).to(self.device), torch.as_tensor(np.asarray(pd, dtype=np.int32)).to(self.device),
                [(q, tuple(ds), lo) for q, ds, lo in groups],
                # what every later call would otherwise recompute: the lists' sizes and offsets, and whether every qid is ONE run of
                # samples (then the predictions are one dict(zip(...)) per query)
                {"counts": counts, "offsets": np.concatenate([[0], np.cumsum(np.asarray(counts, dtype=np.int64))]).astype(np.int64),
                 "one_run_per_qid": len({q for q, _, _ in groups}) == len(groups)})
        q2d = pred_data.qid_to_docids
        # (the lists this rank scores ARE the sampler's lists, docid for docid: what lets later calls compare them element-wise)
        plan[4]["mirrors_q2d"] = plan[4]["one_run_per_qid"] and all(q in q2d and tuple(q2d[q]) == ds for q, ds, _ in plan[3])
        # runs of at least 16 lists and 16,000 pairs are scored in two (from 32 lists and 32,000 pairs: four) parts of about equal size
        # (predict overlaps a part's kernels with the dict building of the parts before it); the pinned buffer the fp16 scores come back
        # in is kept with the plan
        off = plan[4]["offsets"]
        if self.device.type == "cuda" and len(counts) >= 16 and int(off[-1]) >= 16000:
            n_parts = 4 if len(counts) >= 32 and int(off[-1]) >= 32000 else 2
            cuts = sorted({min(max(int(np.searchsorted(off, off[-1] * k // n_parts)), 2), len(counts) - 2) for k in range(1, n_parts)})
            edges = [0] + cuts + [len(counts)]
            parts = [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b - a >= 2]
            if len(parts) >= 2 and parts[0][0] == 0 and parts[-1][1] == len(counts) and all(x[1] == y[0] for x, y in zip(parts[:-1], parts[1:])):
                plan[4]["parts"] = parts
                plan[4]["pinned"] = torch.empty(int(off[-1]), dtype=torch.float16).pin_memory()
        try:
            ref = weakref.ref(pred_data)
        except TypeError:
            ref = (lambda obj: (lambda: obj))(pred_data)
        while len(plans) >= 4:          # a dev set and a test set per rank; old samplers' tables are dropped
            plans.pop(next(iter(plans)))
        plans[key] = (fp, ref, plan)
        return plan

    def _score_store(self, reranker, store, pq, pd, counts, step, offsets=None):
        """Scores index pairs of a candidate store laid out query after query (`counts` documents each) -> fp32 [n].  Rerankers that
        take whole candidate lists get them as lists where the `lists` option allows (every distinct term of a LIST is gathered once:
        csrc/lists.hip); otherwise one launch per `step` pairs."""
        n = int(pq.numel())
        exact = getattr(reranker, "lists_bit_identical", False)
        as_lists = {"never": False, "exact": exact, "always": True}[self.config["lists"]]
        # Which route scores a pair is a function of the reranker and the trainer's configuration - never of how many lists this call
        # (this rank's shard, this part of a run) happens to hold: a reranker whose list scores equal its per-pair scores bit for bit
        # (DRMM, DRMM-TKS, PACRR) may take whichever is faster for the call's shape (a single list, or lists of a few documents, keep the
        # per-pair kernels: one list's passes do not fill the chip); KNRM's pooling sums run in another order on the list route (1e-6
        # relative), so with `lists` = "always" EVERY call of it is a list call, of one list or of a thousand - the fp16 predictions of a
        # query are then the same bits whether it was scored alone, in a 64-query run, or on rank 5 of 8.
        worth_it = len(counts) >= 2 and n >= 8 * len(counts)
        with torch.no_grad():
            if as_lists and len(counts) >= 1 and n >= 1 and (worth_it or not exact) and getattr(reranker, "supports_lists", False) and \
                    store.q_table.shape[1] <= getattr(reranker, "lists_max_qlen", 4):
                if offsets is None:
                    offsets = np.concatenate([[0], np.cumsum(np.asarray(counts, dtype=np.int64))])
                offsets = np.asarray(offsets, dtype=np.int64)
                # the per-pair part of the lists workspace (compact id rows + metadata: 4 L + 32 bytes per pair of the CALL) is bounded by
                # splitting a long run into calls of whole lists (a 1.75 M-pair part at L = 800 asked for 5.6 GB); lists are
                # scored independently, so the scores do not depend on the split
                per_pair = 4 * int(store.d_table.shape[1]) + 32
                if n * per_pair <= self.LISTS_PAIR_BYTES:
                    return reranker.test_resident_lists(store, pq, pd, offsets).float()
                out, a = [], 0
                while a < len(counts):
                    b = a + 1
                    while b < len(counts) and (offsets[b + 1] - offsets[a]) * per_pair <= self.LISTS_PAIR_BYTES:
                        b += 1
                    lo, hi = int(offsets[a]), int(offsets[b])
                    out.append(reranker.test_resident_lists(store, pq[lo:hi], pd[lo:hi], offsets[a:b + 1] - lo).float())
                    a = b
                return torch.cat(out)
            chunks = [reranker.test_resident(store, pq[i:i + step], pd[i:i + step]).float() for i in range(0, n, step)]
        return torch.cat(chunks) if chunks else torch.zeros(0, device=store.device)

    def predict_resident(self, reranker, store, qid_to_docids, pred_fn=None):
        """`predict` over a device-resident `capreolus_amd.feeder.CandidateStore`: no DataLoader,
        no per-batch host->device copy; one kernel launch per `evalbatch` pairs (0 -> the whole run in one launch)."""
        import torch.distributed as dist

        distributed = dist.is_available() and dist.is_initialized()      # (a process group of ONE rank still takes the collective path)
        world = dist.get_world_size() if distributed else 1
        rank = dist.get_rank() if distributed else 0
        reranker.model.to(store.device).eval()
        qids = list(qid_to_docids.keys())
        b = shard_bounds([len(qid_to_docids[q]) for q in qids], world)
        mine = {q: qid_to_docids[q] for q in qids[b[rank]:b[rank + 1]]}
        keys, pq, pd = store.pairs(mine)
        step = self.config["evalbatch"] if self.config["evalbatch"] > 0 else max(len(keys), 1)
        local = self._score_store(reranker, store, pq, pd, [len(v) for v in mine.values()], step)
        if distributed:
            counts = [sum(len(qid_to_docids[q]) for q in qids[b[r]:b[r + 1]]) for r in range(world)]
            width = max(counts)
            padded = torch.zeros(width, dtype=torch.float32, device=store.device)
            padded[: local.numel()] = local
            gathered = torch.empty(width * world, dtype=torch.float32, device=store.device)
            dist.all_gather_into_tensor(gathered, padded)
            g = gathered.cpu().numpy().reshape(world, width)
            allscores = np.concatenate([g[r, : counts[r]] for r in range(world)])
            allkeys = [(q, d) for q in qids for d in qid_to_docids[q]]
        else:
            allkeys, allscores = keys, local.cpu().numpy()
        preds = _preds_dict(allkeys, allscores)
        if pred_fn is not None and rank == 0:
            os.makedirs(os.path.dirname(os.fspath(pred_fn)) or ".", exist_ok=True)
            write_trec_run(preds, pred_fn)
        return preds

    def evaluate_resident(self, reranker, store, qid_to_docids, qrels, k=20):
        """Dev-set nDCG@k of `reranker` over a device-resident candidate store without bringing the scores to the host
        (SURVEY.md §8f row N2): scoring kernels -> `capamd_ndcg_cut`; only one fp64 per query is copied back.  The value
        equals `evaluator.eval_runs(predict(...), qrels, ["ndcg_cut_k"])` of the reference (mean over the queries that have
        qrels; scores rounded to fp16 first, as `predict` stores them)."""
        from .. import ranking

        reranker.model.to(store.device).eval()
        # The index pairs and the judgment arrays depend only on (store, run, qrels, k): `train` evaluates the same dev set after every
        # iteration, so they are built once (walking 64,000 dict entries costs 10x the scoring and ranking kernels) and kept while the
        # caller holds the same objects.
        # (+ a cheap fingerprint of their sizes, so that a run or judgments grown / shrunk in place are noticed; values edited in place
        # under an unchanged shape are not - hand in new objects then)
        key = (id(store), id(qid_to_docids), id(qrels), k, len(qid_to_docids), sum(len(v) for v in qid_to_docids.values()),
               len(qrels), sum(len(v) for v in qrels.values()))
        plan = getattr(self, "_eval_plan", None)
        if plan is None or plan[0] != key or plan[1] is not store or plan[2] is not qid_to_docids or plan[3] is not qrels:
            keys, pq, pd = store.pairs(qid_to_docids)
            rel, tie, idcg, offsets = ranking.eval_arrays(qid_to_docids, qrels, k, store.device)
            judged = torch.tensor([q in qrels for q in qid_to_docids], dtype=torch.bool, device=store.device)
            counts = [len(v) for v in qid_to_docids.values()]
            plan = self._eval_plan = (key, store, qid_to_docids, qrels, len(keys), pq, pd, rel, tie, idcg, offsets, judged, counts)
        n, pq, pd, rel, tie, idcg, offsets, judged, counts = plan[4:]
        step = self.config["evalbatch"] if self.config["evalbatch"] > 0 else max(n, 1)
        scores = self._score_store(reranker, store, pq, pd, counts, step)
        per_query = ranking.ndcg_cut(scores, offsets, rel, tie, idcg, k=k)
        from ..engine import status_word

        status_word(store.device).raise_if_set()
        return float(per_query[judged].mean().item()) if bool(judged.any()) else 0.0

    def predict(self, reranker, pred_data, pred_fn=None):
        """Scores every (qid, docid) of `pred_data`; returns {qid: {docid: score}} on every rank and
        writes the TREC run to `pred_fn` (rank 0)."""
        import torch.distributed as dist

        distributed = dist.is_available() and dist.is_initialized()      # (a process group of ONE rank still takes the collective path)
        world = dist.get_world_size() if distributed else 1
        rank = dist.get_rank() if distributed else 0
        if torch.cuda.is_available():
            self.device = torch.device("cuda", torch.cuda.current_device())
        else:
            self.device = torch.device("cpu")
        model = reranker.model
        first = next(model.parameters(), None)
        if first is None or first.device != self.device:      # (nn.Module.to walks every parameter even when there is nothing to move)
            model = model.to(self.device)
        if model.training:
            model.eval()

        part, offset, count, total = shard_pred_data(pred_data, rank, world)
        evalbatch = self.config["evalbatch"] if self.config["evalbatch"] > 0 else self.config["batch"]
        workers = 1 if self.config["multithread"] else 0
        keys, chunks = [], []
        plan = None
        if count > 0 and self.config["resident"] and self.device.type == "cuda" and getattr(reranker, "supports_resident", False):
            plan = self._resident_plan(pred_data, part, rank, world)
        if plan is not None:
            from .. import engine, pyhost

            store, pq, pd, groups, extra = plan
            step = max(evalbatch, self.config["coalesce"])
            # (a run of one qid = one list; its query row is the same for every pair, checked when the plan was built)
            # One synchronisation per call: the kernels' status word is read after the scores have come back (deferred_status), and the
            # reference's `score.astype(np.float16)` (trainer/pytorch.py:346-348; round to nearest even) runs on the device, so that the
            # copy is 2 bytes per pair and `tolist` is all the host still does per score.
            parts = extra.get("parts") if not distributed and extra["one_run_per_qid"] else None
            if parts:
                # The lists in two or four parts (round 4): a part's kernels run while the host turns the parts before it into dicts -
                # 64,000 dict inserts and float objects cost CPython twice what the kernels take.  Same kernels, same scores.
                with engine.deferred_status(self.device):
                    host = extra["pinned"]
                    done = []
                    for g0, g1 in parts:
                        lo, hi = int(extra["offsets"][g0]), int(extra["offsets"][g1])
                        sc = self._score_store(reranker, store, pq[lo:hi], pd[lo:hi], extra["counts"][g0:g1], step, extra["offsets"][g0:g1 + 1] - lo)
                        host[lo:hi].copy_(sc.to(torch.float16), non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record()
                        done.append(ev)
                    preds, arr = {}, host.numpy()
                    for (g0, g1), ev in zip(parts, done):
                        ev.synchronize()
                        pyhost.preds_from_fp16(groups, arr, preds, g0, g1)      # preds[qid] = dict(zip(docids, that list's fp16 scores))
                if pred_fn is not None:
                    os.makedirs(os.path.dirname(os.fspath(pred_fn)) or ".", exist_ok=True)
                    write_trec_run(preds, pred_fn)
                return preds
            with engine.deferred_status(self.device):
                chunks = [self._score_store(reranker, store, pq, pd, extra["counts"], step, extra["offsets"])]
                if not distributed:      # the {qid: {docid: score}} dict straight from the per-query slices
                    vals = np.ascontiguousarray(chunks[0].to(torch.float16).cpu().numpy())
            if not distributed:
                if len(vals) != count:
                    raise RuntimeError(f"rank {rank} scored {len(vals)} pairs, expected {count}")
                preds = pyhost.preds_from_fp16(groups, vals, {}, merge=not extra["one_run_per_qid"])
                if pred_fn is not None:
                    os.makedirs(os.path.dirname(os.fspath(pred_fn)) or ".", exist_ok=True)
                    write_trec_run(preds, pred_fn)
                return preds
            keys = [(q, d) for q, ds, _ in groups for d in ds]
        elif count > 0:
            # (no pin_memory, unlike trainer/pytorch.py:335: on ROCm the loader's pinning thread allocates a fresh pinned block for every
            # batch of more than ~1 MB - measured 1.29 s against 0.08 s per 20,000 samples at evalbatch 256, scripts/dbg/pin_probe.py - and
            # at the default 32 the pageable copy is faster too, 0.098 against 0.124 s)
            loader = torch.utils.data.DataLoader(part, batch_size=evalbatch, pin_memory=False,
                                                 num_workers=workers)
            coalesce = 0 if getattr(reranker, "batch_coupled", False) else self.config["coalesce"]
            pending, n_pending = [], 0
            byte_cap = [None]       # pairs per merged batch allowed by `coalesce_bytes`, known after the first batch

            def score(batch, n):
                dbatch = {k: v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v for k, v in batch.items()}
                scores = reranker.test(dbatch).view(-1)[:n]
                chunks.append(scores.float())      # stays on the device: no per-batch sync
                keys.extend(zip(batch["qid"][:n], batch["posdocid"][:n]))

            def flush():
                if not pending:
                    return
                if len(pending) == 1:
                    merged = pending[0]
                else:
                    merged = {k: (torch.cat([b[k] for b in pending]) if torch.is_tensor(v) else
                                  np.concatenate([b[k] for b in pending]) if isinstance(v, np.ndarray) else
                                  [x for b in pending for x in b[k]])
                              for k, v in pending[0].items()}
                score(merged, len(merged["qid"]))
                pending.clear()

            with torch.no_grad():
                for batch in loader:
                    n = len(batch["qid"])
                    if coalesce > 0:
                        if byte_cap[0] is None:
                            per_pair = sum(v.numel() * v.element_size() for v in batch.values() if torch.is_tensor(v)) / max(n, 1)
                            byte_cap[0] = max(evalbatch, int(self.config["coalesce_bytes"] // max(per_pair, 1)))
                        pending.append(batch)
                        n_pending += n
                        if n_pending >= min(coalesce, byte_cap[0]):
                            flush()
                            n_pending = 0
                        continue
                    if n != evalbatch:
                        batch = self.fill_incomplete_batch(batch, batch_size=evalbatch)
                    score(batch, n)
                flush()
        local = torch.cat(chunks) if chunks else torch.zeros(0, device=self.device)
        if local.numel() != count:
            raise RuntimeError(f"rank {rank} scored {local.numel()} pairs, expected {count}")

        if distributed:
            # every rank can derive every rank's (offset, count) from the sampler; only scores travel
            plan = [shard_pred_data(pred_data, r, world)[1:3] for r in range(world)]
            width = max(c for _, c in plan)
            padded = torch.zeros(width, dtype=torch.float32, device=self.device)
            padded[:count] = local
            gathered = torch.empty(width * world, dtype=torch.float32, device=self.device)
            dist.all_gather_into_tensor(gathered, padded)   # the one collective of the path (RCCL over xGMI)
            gathered = gathered.cpu().numpy().reshape(world, width)
            allscores = np.concatenate([gathered[r, :c] for r, (_, c) in enumerate(plan)])
            if hasattr(pred_data, "get_qid_docid_pairs"):   # reference PredSampler, sampler/__init__.py:257-264
                allkeys = list(pred_data.get_qid_docid_pairs())
            else:                                            # opaque sampler: ids are only known where they were read
                parts = [None] * world
                dist.all_gather_object(parts, keys)
                allkeys = [k for ks in parts for k in ks]
            if len(allkeys) != len(allscores):
                raise RuntimeError(f"gathered {len(allscores)} scores for {len(allkeys)} (qid, docid) pairs")
        else:
            allkeys, allscores = keys, local.cpu().numpy()

        preds = _preds_dict(allkeys, allscores)
        if pred_fn is not None and rank == 0:
            os.makedirs(os.path.dirname(os.fspath(pred_fn)) or ".", exist_ok=True)
            write_trec_run(preds, pred_fn)
        return preds
