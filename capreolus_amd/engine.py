"""Thin PyTorch plumbing over the C-ABI library: device memory, streams, error surfacing.

PyTorch is used for exactly three things here: owning device buffers (caching allocator),
naming the current HIP stream, and holding the model parameters.  All arithmetic of the
scoring path happens inside libcapreolus_amd.so (capreolus_amd/csrc/*.hip).  There is no CPU
or eager-PyTorch fallback: inputs that are not on a HIP device raise.
"""
import ctypes
import math
import os

import threading

import torch

from . import _lib

HIST_TYPES = {"CH": 0, "NH": 1, "LCH": 2}
GATE_TYPES = {"IDF": 0, "TV": 1}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "capreolus_amd scores on an MI355X only: got a tensor on %s. There is no CPU fallback; "
                "move the batch and the model to 'cuda' (PytorchTrainer.predict does)." % t.device
            )


def _i64(t):
    return t if (t.dtype == torch.int64 and t.is_contiguous()) else t.to(torch.int64).contiguous()


def _f32(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


_deferred = [0]


class deferred_status:
    """Inside this context the per-call status read-backs (one 4-byte device->host copy, i.e. one synchronisation, per scoring
    call) are skipped; the bits keep accumulating in the device word and are raised when the context exits.  For callers that
    queue many scoring calls back to back (bench.py, a resident evaluation loop)."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())

    def __enter__(self):
        _deferred[0] += 1
        return self

    def __exit__(self, *exc):
        _deferred[0] -= 1
        if exc[0] is None and _deferred[0] == 0:
            status_word(self.device).raise_if_set()
        return False


class StatusWord:
    """The device int32 the kernels OR data-dependent error bits into (include/capreolus_amd.h)."""

    def __init__(self, device):
        self.t = torch.zeros(1, dtype=torch.int32, device=device)

    def raise_if_set(self):
        if _deferred[0]:
            return
        bits = int(self.t.item())  # synchronises, like the reference's .cpu() (trainer/pytorch.py:345)
        if bits:
            self.t.zero_()
            msgs = []
            if bits & _lib.STATUS_DOC_ID_RANGE:
                msgs.append("a document term id is outside the embedding table")
            if bits & _lib.STATUS_QUERY_ID_RANGE:
                msgs.append("a query term id is >= the embedding table size")
            if bits & _lib.STATUS_QUERY_OOV:
                msgs.append("a negative (OOV) query term id where the reference model cannot take one (DRMM.py:109, nn.Embedding in ConvKNRM.py:43)")
            if bits & _lib.STATUS_LIST_QUERY:
                raise ValueError("a candidate list holds pairs of more than one query (or idf row): a list is the pairs of ONE query, "
                                 "laid out as its offsets say (capamd_*_forward_lists)")
            if bits & (_lib.STATUS_SCORE_NAN | _lib.STATUS_TIE_RANGE):
                raise ValueError("ranking: " + ("a score is NaN; " if bits & _lib.STATUS_SCORE_NAN else "") +
                                 ("a tie-break rank is outside 0..n-1" if bits & _lib.STATUS_TIE_RANGE else ""))
            raise IndexError("index out of range in self: " + "; ".join(msgs))


_status_words = {}


def status_word(device):
    key = (device.type, device.index)
    if key not in _status_words:
        _status_words[key] = StatusWord(device)
    return _status_words[key]


_launch_hint = threading.local()


class concurrent_launches:
    """Context: the caller keeps several scoring calls in flight on different streams (the CAPAMD_LAUNCH_CONCURRENT flag of the
    interaction entries), e.g. one candidate list per launch round-robin over a few streams so that the tail of one list overlaps
    the next launch.  Per thread and nestable; the flag travels with each call, the library keeps no state."""

    def __enter__(self):
        _launch_hint.depth = getattr(_launch_hint, "depth", 0) + 1
        return self

    def __exit__(self, *exc):
        _launch_hint.depth -= 1
        return False


def _launch_flags():
    return _lib.LAUNCH_CONCURRENT if getattr(_launch_hint, "depth", 0) else 0


_workspaces = {}


def _workspace(device):
    """The few bytes of device memory an interaction call may use (capamd_interaction_workspace_bytes): one per (device, stream) -
    calls on one stream run in order, calls on different streams must not share it."""
    key = (device.index, int(torch.cuda.current_stream(device).cuda_stream))
    ws = _workspaces.get(key)
    if ws is None:
        ws = _workspaces[key] = torch.zeros(max(16, int(_lib.load().capamd_interaction_workspace_bytes()) // 4), dtype=torch.int32, device=device)
    return ws


class PackedEmbedding:
    """The embedding table re-laid out for the gather kernels (capamd_pack_embeddings).

    Re-packed whenever the source weight changes (tracked with the tensor's version counter and
    storage pointer), so `load_weights` / fine-tuning never score against stale rows.
    """

    def __init__(self):
        self._key = None
        self.packed = None
        self.V = self.D = 0

    def get(self, weight):
        _need_gpu(weight)
        key = (weight.data_ptr(), weight._version, tuple(weight.shape), weight.device.index)
        if key != self._key:
            lib = _lib.load()
            w = _f32(weight.detach())
            V, D = w.shape
            nbytes = lib.capamd_packed_table_bytes(V, D)
            if nbytes < 0:
                raise ValueError(f"embedding dimension {D} is not supported by the packed layout (D <= 319)")
            packed = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
            _lib.check(lib.capamd_pack_embeddings(_ptr(w), V, D, w.stride(0), _ptr(packed), _stream()), "capamd_pack_embeddings")
            self.packed, self.V, self.D, self._key = packed, V, D, key
        return self.packed


def similarity_matrix(query, doc, packed, V, D, check=True):
    """SimilarityMatrix.forward (reference common.py:170-182): fp32 [B, Q, L]."""
    _need_gpu(query, doc, packed)
    q, d = _i64(query), _i64(doc)
    B, Q = q.shape
    L = d.shape[1]
    out = torch.empty((B, Q, L), dtype=torch.float32, device=q.device)
    st = status_word(q.device)
    rc = _lib.load().capamd_similarity_matrix(_ptr(q), _ptr(d), B, Q, L, _ptr(packed), V, D, _ptr(out), _ptr(st.t), _stream())
    _lib.check(rc, "capamd_similarity_matrix")
    if check:
        st.raise_if_set()
    return out


def knrm_forward(query, doc, packed, V, D, mu, sigma, w1, b1, w2=None, b2=None, scoretanh=False, out=None, check=True):
    """KNRM_class.forward (reference KNRM.py:39-55): fp32 [B]."""
    _need_gpu(query, doc, packed, mu, sigma, w1, b1, w2, b2)
    q, d = _i64(query), _i64(doc)
    B, Q = q.shape
    L = d.shape[1]
    if d.shape[0] != B:
        raise AssertionError("query and document batch sizes differ")  # common.py:172
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=q.device)
    hidden = 0 if w2 is None else w1.shape[0]
    st, ws = status_word(q.device), _workspace(q.device)
    rc = _lib.load().capamd_knrm_forward(
        _ptr(q), _ptr(d), B, Q, L, _ptr(packed), V, D, _ptr(mu), _ptr(sigma), mu.numel(), _ptr(w1), _ptr(b1), hidden,
        _ptr(w2), _ptr(b2), int(bool(scoretanh)), _ptr(out), _ptr(st.t), _ptr(ws), ws.numel() * 4, _launch_flags(), _stream())
    _lib.check(rc, "capamd_knrm_forward")
    if check:
        st.raise_if_set()
    return out


def drmm_forward(query, doc, idf, packed, V, D, edges, hist_type, gate_type, gate_w, emb_raw, w1, b1, w2, b2, out_w, out_b,
                 out=None, counts_out=None, check=True):
    """DRMM_class.forward (reference DRMM.py:101-116): fp32 [B]."""
    _need_gpu(query, doc, idf, packed, edges, gate_w, w1, b1, w2, b2, out_w, out_b)
    q, d, idf = _i64(query), _i64(doc), _f32(idf)
    B, Q = q.shape
    L = d.shape[1]
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=q.device)
    st, ws = status_word(q.device), _workspace(q.device)
    ld = emb_raw.stride(0) if emb_raw is not None else 0
    rc = _lib.load().capamd_drmm_forward(
        _ptr(q), _ptr(d), _ptr(idf), B, Q, L, _ptr(packed), V, D, _ptr(edges), edges.numel(), HIST_TYPES[hist_type],
        GATE_TYPES[gate_type], _ptr(gate_w), _ptr(emb_raw), ld, _ptr(w1), _ptr(b1), w1.shape[0], _ptr(w2), _ptr(b2),
        _ptr(out_w), _ptr(out_b), _ptr(out), _ptr(counts_out), _ptr(st.t), _ptr(ws), ws.numel() * 4, _launch_flags(), _stream())
    _lib.check(rc, "capamd_drmm_forward")
    if check:
        st.raise_if_set()
    return out


AGGREGATIONS = {"max": 0, "first": 1, "sum": 2, "avg": 3}

_LAYER_TENSORS = (
    "attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight", "attention.self.key.bias",
    "attention.self.value.weight", "attention.self.value.bias", "attention.output.dense.weight", "attention.output.dense.bias",
    "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias", "intermediate.dense.weight", "intermediate.dense.bias",
    "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias",
)


SUPPORTED_LENGTHS = tuple(range(32, 257, 32)) + (384, 512)   # the attention kernels of the library (capamd_bert_maxp_forward)


class BertEngine:
    """Device-side state of one BERT sequence classifier for capamd_bert_maxp_forward.

    `params` maps the HF state_dict names (``bert.embeddings...``, ``bert.encoder.layer.N...``,
    ``bert.pooler.dense...``, ``classifier...``) to live fp32 parameters on the device.  The bf16
    weight blob is rebuilt whenever any parameter's version counter or storage changes.
    """

    COMPUTE_DTYPES = {"bf16": 0, "fp16": 1}

    def __init__(self, params, heads, microbatch=256, compute_dtype="fp16", skip_padding=True, ln_eps=0.0, pos_pad_id=-1):
        """ln_eps / pos_pad_id: RoBERTa bodies (include/capreolus_amd.h: capamd_bert_model) - LayerNorm epsilon (0 = BERT's 1e-12) and the
        padding id its position ids are counted around (-1 = BERT: position i)."""
        if compute_dtype not in self.COMPUTE_DTYPES:
            raise ValueError("compute_dtype must be 'bf16' or 'fp16'")
        self.ln_eps, self.pos_pad_id = float(ln_eps), int(pos_pad_id)
        self.params = params
        self.heads = heads
        self.microbatch = microbatch
        self.compute_dtype = compute_dtype
        self.skip_padding = skip_padding
        self._key = None
        self._blob = self._lf32 = None
        self._wss = {}        # workspaces: None -> the full-length call, Sb -> the length bucket running on its own stream
        self._streams = {}
        # the full-length path splits a large batch over this many HIP streams (own workspaces); 1 = strictly serial kernels
        self.n_streams = max(1, int(os.environ.get("CAPAMD_BERT_STREAMS", "2")))
        self._model = None
        self._keep = None

    def _dims(self):
        p = self.params
        hidden = p["bert.embeddings.word_embeddings.weight"].shape[1]
        layers = 1 + max(int(k.split(".")[3]) for k in p if k.startswith("bert.encoder.layer."))
        m = _lib.BertModel()
        m.hidden, m.layers, m.heads = hidden, layers, self.heads
        m.ffn = p["bert.encoder.layer.0.intermediate.dense.weight"].shape[0]
        m.vocab = p["bert.embeddings.word_embeddings.weight"].shape[0]
        m.max_pos = p["bert.embeddings.position_embeddings.weight"].shape[0]
        m.type_vocab = p["bert.embeddings.token_type_embeddings.weight"].shape[0]
        m.compute_dtype = self.COMPUTE_DTYPES[self.compute_dtype]
        m.ln_eps, m.pos_pad_id = self.ln_eps, self.pos_pad_id
        return m

    def model(self):
        p = self.params
        key = tuple((t.data_ptr(), t._version) for t in p.values()) + (self.compute_dtype,)
        if key == self._key:
            return self._model
        lib = _lib.load()
        some = p["bert.embeddings.word_embeddings.weight"]
        _need_gpu(*p.values())
        m = self._dims()
        nblob, nf = lib.capamd_bert_blob_bytes(ctypes.byref(m)), lib.capamd_bert_layer_f32_floats(ctypes.byref(m))
        if nblob < 0:
            raise ValueError("unsupported BERT geometry: need hidden == 64*heads, hidden % 64 == 0 (<= 1024), ffn % 64 == 0")
        blob = torch.empty(nblob, dtype=torch.uint8, device=some.device)
        lf32 = torch.empty(nf * m.layers, dtype=torch.float32, device=some.device)
        keep = {k: _f32(v.detach()) for k, v in p.items()}
        for layer in range(m.layers):
            ptrs = [keep[f"bert.encoder.layer.{layer}.{n}"].data_ptr() for n in _LAYER_TENSORS]
            # + the LayerNorm whose output feeds this layer (the previous layer's output LayerNorm; layer 0 reads the
            # already normalised embeddings): folded into this layer's QKV / O-proj epilogues (bert_gemm.h)
            prev = [keep[f"bert.encoder.layer.{layer - 1}.output.LayerNorm.{n}"].data_ptr() for n in ("weight", "bias")] if layer > 0 else [None, None]
            arr = (ctypes.c_void_p * 18)(*(ptrs + prev))
            _lib.check(lib.capamd_bert_pack_layer(ctypes.byref(m), layer, arr, _ptr(blob), _ptr(lf32), _stream()), "capamd_bert_pack_layer")
        for field, name in (("word_emb", "bert.embeddings.word_embeddings.weight"), ("pos_emb", "bert.embeddings.position_embeddings.weight"),
                            ("type_emb", "bert.embeddings.token_type_embeddings.weight"), ("emb_ln_g", "bert.embeddings.LayerNorm.weight"),
                            ("emb_ln_b", "bert.embeddings.LayerNorm.bias"), ("pooler_w", "bert.pooler.dense.weight"),
                            ("pooler_b", "bert.pooler.dense.bias"), ("cls_w", "classifier.weight"), ("cls_b", "classifier.bias")):
            # (pooler / classifier are absent for encoders that are only tapped - CEDR-KNRM on an ELECTRA body; capamd_bert_maxp_forward
            # refuses a model without them)
            setattr(m, field, keep[name].data_ptr() if name in keep else None)
        m.blob, m.layer_f32 = blob.data_ptr(), lf32.data_ptr()
        self._blob, self._lf32, self._keep, self._model, self._key = blob, lf32, keep, m, key
        return m

    def _encode(self, ids, mask, seg, B, P, S, aggregation, out, plog, check, ws_key=None):
        """One capamd_bert_maxp_forward call over [B, P, S] (all passages at length S) on the current stream; `ws_key` selects
        the workspace (one per concurrently running stream)."""
        m = self.model()
        lib = _lib.load()
        # `microbatch` counts passages of 256 tokens: shorter passages go in proportionally larger micro-batches (same rows, same
        # workspace, and the GEMMs of a short-passage bucket still see 65,536 rows)
        mb = min(self.microbatch * max(1, 256 // S), B * P)
        need = lib.capamd_bert_workspace_bytes(ctypes.byref(m), S, mb, B * P)
        if need < 0:
            raise ValueError(f"unsupported passage length {S} (supported: multiples of 32 up to 256, 384, 512)")
        ws = self._wss.get(ws_key)
        if ws is None or ws.numel() < need or ws.device != ids.device:
            ws = self._wss[ws_key] = torch.empty(need, dtype=torch.uint8, device=ids.device)
        st = status_word(ids.device)
        rc = lib.capamd_bert_maxp_forward(_ptr(ids), _ptr(mask), _ptr(seg), B, P, S, ctypes.byref(m), AGGREGATIONS[aggregation], mb,
                                          _ptr(ws), ws.numel(), _ptr(out), _ptr(plog), _ptr(st.t), _stream())
        _lib.check(rc, "capamd_bert_maxp_forward")
        if check:
            st.raise_if_set()

    def forward(self, doc_input, doc_mask, doc_seg, aggregation="max", return_passage_logits=False, check=True, skip_padding=None):
        """PTBERTMaxP_Class.predict_step (reference ptBERTMaxP.py:67-96): int64 [B,P,S] x3 -> fp32 [B].

        skip_padding (default: the engine's setting): encode every passage at the shortest supported length (SUPPORTED_LENGTHS) that
        holds its last attended token instead of at S.  Passages are independent, padded key positions get an attention
        weight of exactly 0 and padded query positions never reach the [CLS] row, so the passage logits are bit-identical to
        the full-length computation; only the dead rows are not computed.  Costs one small device->host copy (the bucket
        sizes) per call, like the reference's own `.cpu()` per batch."""
        _need_gpu(doc_input, doc_mask, doc_seg)
        if aggregation not in AGGREGATIONS:
            raise ValueError("Unknown aggregation method: {}".format(aggregation))
        ids, mask, seg = _i64(doc_input), _i64(doc_mask), _i64(doc_seg)
        B, P, S = ids.shape
        out = torch.empty(B, dtype=torch.float32, device=ids.device)
        if B == 0:
            return (out, torch.empty(0, device=ids.device)) if return_passage_logits else out
        if skip_padding is None:
            skip_padding = self.skip_padding
        if aggregation == "first" and not return_passage_logits and P > 1 and skip_padding:
            # ptBERTMaxP.py:85-86 takes scores[:, 0]: the other passages of a document are never read
            return self.forward(ids[:, :1], mask[:, :1], seg[:, :1], "first", False, check, skip_padding)
        lengths = [x for x in SUPPORTED_LENGTHS if x < S] if (skip_padding and S in SUPPORTED_LENGTHS) else []
        if not lengths:
            NP = B * P
            q = 256 // math.gcd(S, 256)                      # passages per whole 256-row GEMM tile
            ns = max(1, min(self.n_streams, NP // self.microbatch))
            per = (NP // ns + q - 1) // q * q if ns > 1 else NP   # passages per stream, in whole tiles
            if ns > 1 and NP // ns >= 4 * self.microbatch and S == 256:
                # long runs: cut at whole micro-batches, so that every stream but the last runs FULL micro-batches only (whole rounds of
                # GEMM tiles on every CU, bert.hip: plan_microbatch) and one tail exists per call, not one per stream
                per = (NP // ns + self.microbatch - 1) // self.microbatch * self.microbatch
                if (ns - 1) * per >= NP:      # (five or more streams: rounding UP leaves the last stream nothing - round down instead)
                    per = NP // ns // self.microbatch * self.microbatch
            cuts = [min(NP, k * per) for k in range(ns)] + [NP]
            if ns < 2 or cuts[-2] >= NP:
                plog = torch.empty(NP, dtype=torch.float32, device=ids.device) if return_passage_logits else None
                self._encode(ids, mask, seg, B, P, S, aggregation, out, plog, check)
                return (out, plog) if return_passage_logits else out
            # full-length computation of a large batch: `ns` slices on as many streams (own workspaces) - the persistent GEMM kernels
            # of one slice start on the CUs the others' last tiles, low-occupancy kernels and launch gaps leave idle
            fids, fmask, fseg = ids.view(NP, S), mask.view(NP, S), seg.view(NP, S)
            plog = torch.empty(NP, dtype=torch.float32, device=ids.device)
            self.model()
            main = torch.cuda.current_stream(ids.device)
            sides = []
            for k, (a0, a1) in enumerate(zip(cuts[:-1], cuts[1:])):
                side = self._streams.get(("half", k))
                if side is None:
                    side = self._streams[("half", k)] = torch.cuda.Stream(device=ids.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._encode(fids[a0:a1], fmask[a0:a1], fseg[a0:a1], a1 - a0, 1, S, "first", plog[a0:a1], None, False, ws_key=("half", k))
                sides.append(side)
            for side in sides:
                main.wait_stream(side)
            if check:
                status_word(ids.device).raise_if_set()
            cnt = torch.empty(1, dtype=torch.int32, device=ids.device)
            _lib.check(_lib.load().capamd_maxp_pool(_ptr(plog), _ptr(mask), _ptr(seg), B, P, S, AGGREGATIONS[aggregation], _ptr(out), _ptr(cnt),
                                                    _stream()), "capamd_maxp_pool")
            return (out, plog) if return_passage_logits else out

        NP = B * P
        fids, fmask, fseg = ids.view(NP, S), mask.view(NP, S), seg.view(NP, S)
        # position after the last attended token of each passage (masks with holes keep their full extent)
        end = ((fmask != 0) * torch.arange(1, S + 1, device=ids.device)).amax(dim=1)
        bounds = lengths + [S]
        bucket = torch.bucketize(end, torch.tensor(lengths, device=ids.device), right=False)   # 0: <= lengths[0], ...
        order = torch.argsort(bucket, stable=True)
        counts = torch.bincount(bucket, minlength=len(bounds)).cpu().tolist()   # the one host round trip of this call
        plog = torch.empty(NP, dtype=torch.float32, device=ids.device)
        self.model()                                  # (re)pack on the caller's stream before the buckets fan out
        main = torch.cuda.current_stream(ids.device)
        lo = 0
        used = []
        for Sb, n in zip(bounds, counts):
            if n == 0:
                continue
            sel = order[lo:lo + n]
            lo += n
            # the buckets are independent: each runs on its own stream with its own workspace, so one bucket's kernels fill the
            # CUs another bucket's last tiles leave idle and the launch gaps of one hide behind the work of the others
            side = self._streams.get(Sb)
            if side is None:
                side = self._streams[Sb] = torch.cuda.Stream(device=ids.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                # (a ragged bucket is padded to whole 256-row GEMM tiles inside capamd_bert_maxp_forward)
                bi, bm, bs = (t.index_select(0, sel)[:, :Sb].contiguous() for t in (fids, fmask, fseg))
                o = torch.empty(n, dtype=torch.float32, device=ids.device)
                self._encode(bi, bm, bs, n, 1, Sb, "first", o, None, False, ws_key=Sb)
                plog.index_copy_(0, sel, o)
                for t in (bi, bm, bs, o, sel):
                    t.record_stream(side)
            used.append(side)
        for side in used:
            main.wait_stream(side)
        if check:
            status_word(ids.device).raise_if_set()
        cnt = torch.empty(1, dtype=torch.int32, device=ids.device)
        _lib.check(_lib.load().capamd_maxp_pool(_ptr(plog), _ptr(mask), _ptr(seg), B, P, S, AGGREGATIONS[aggregation], _ptr(out), _ptr(cnt),
                                                _stream()), "capamd_maxp_pool")
        return (out, plog) if return_passage_logits else out


def _i32(t):
    return t if (t.dtype == torch.int32 and t.is_contiguous()) else t.to(torch.int32).contiguous()


def knrm_forward_indexed(q_table, d_table, pair_q, pair_d, packed, V, D, mu, sigma, w1, b1, w2=None, b2=None, scoretanh=False,
                         out=None, check=True):
    """KNRM over a device-resident candidate store (capamd_knrm_forward_indexed): int32 tables + per-pair rows."""
    _need_gpu(q_table, d_table, pair_q, pair_d, packed, mu, sigma, w1, b1, w2, b2)
    qt, dt, pq, pd = _i32(q_table), _i32(d_table), _i32(pair_q), _i32(pair_d)
    B, Q, L = pq.numel(), qt.shape[1], dt.shape[1]
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=pq.device)
    hidden = 0 if w2 is None else w1.shape[0]
    st, ws = status_word(pq.device), _workspace(pq.device)
    rc = _lib.load().capamd_knrm_forward_indexed(
        _ptr(qt), _ptr(dt), _ptr(pq), _ptr(pd), B, Q, L, _ptr(packed), V, D, _ptr(mu), _ptr(sigma), mu.numel(), _ptr(w1), _ptr(b1),
        hidden, _ptr(w2), _ptr(b2), int(bool(scoretanh)), _ptr(out), _ptr(st.t), _ptr(ws), ws.numel() * 4, _launch_flags(), _stream())
    _lib.check(rc, "capamd_knrm_forward_indexed")
    if check:
        st.raise_if_set()
    return out


def drmm_forward_indexed(q_table, d_table, idf_table, pair_q, pair_d, packed, V, D, edges, hist_type, gate_type, gate_w, emb_raw,
                         w1, b1, w2, b2, out_w, out_b, out=None, check=True):
    """DRMM over a device-resident candidate store (capamd_drmm_forward_indexed)."""
    _need_gpu(q_table, d_table, idf_table, pair_q, pair_d, packed, edges, gate_w, w1, b1, w2, b2, out_w, out_b)
    qt, dt, pq, pd, idf = _i32(q_table), _i32(d_table), _i32(pair_q), _i32(pair_d), _f32(idf_table)
    B, Q, L = pq.numel(), qt.shape[1], dt.shape[1]
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=pq.device)
    st, ws = status_word(pq.device), _workspace(pq.device)
    ld = emb_raw.stride(0) if emb_raw is not None else 0
    rc = _lib.load().capamd_drmm_forward_indexed(
        _ptr(qt), _ptr(dt), _ptr(idf), _ptr(pq), _ptr(pd), B, Q, L, _ptr(packed), V, D, _ptr(edges), edges.numel(),
        HIST_TYPES[hist_type], GATE_TYPES[gate_type], _ptr(gate_w), _ptr(emb_raw), ld, _ptr(w1), _ptr(b1), w1.shape[0], _ptr(w2),
        _ptr(b2), _ptr(out_w), _ptr(out_b), _ptr(out), None, _ptr(st.t), _ptr(ws), ws.numel() * 4, _launch_flags(), _stream())
    _lib.check(rc, "capamd_drmm_forward_indexed")
    if check:
        st.raise_if_set()
    return out


# ---- whole candidate lists (capamd_*_forward_lists) ------------------------------------------------------------------------------
_list_workspaces = {}
# Bytes the per-list part of a whole-list call's workspace may take (17 B x V per list in flight: 6.8 MB at V = 400,001, but 68 MB at
# V = 4 M - 64 lists would be 4.3 GB, the 256 a launch group takes 17 GB): fewer lists are kept in flight when it would be exceeded (the library then works through the
# lists in more, smaller groups).  The per-pair part (4 L + 32 bytes per pair of the call) comes on top.
LISTS_WORKSPACE_BUDGET = 2 << 30
LISTS_MAX_QLEN = 8          # query terms a whole-list call takes (csrc/lists.h: kListMaxQ; PACRR: 4)


def _lists_workspace(device, n_lists, V, n_pairs, L, Q=4):
    lib = _lib.load()
    one = int(lib.capamd_lists_workspace_bytes_q(1, int(V), 0, int(L), int(Q)))
    per_list = int(lib.capamd_lists_workspace_bytes_q(2, int(V), 0, int(L), int(Q))) - one
    in_flight = max(1, min(int(n_lists), LISTS_WORKSPACE_BUDGET // max(per_list, 1)))
    nbytes = int(lib.capamd_lists_workspace_bytes_q(in_flight, int(V), int(n_pairs), int(L), int(Q)))
    if nbytes == 0:
        raise ValueError(f"whole-list scoring takes queries of up to {LISTS_MAX_QLEN} terms, got {Q}")
    key = (device.index, int(torch.cuda.current_stream(device).cuda_stream))
    ws = _list_workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _list_workspaces[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ws[:nbytes]


def release_workspaces():
    """Drops the cached whole-list workspaces (one per device and stream; a later call allocates again)."""
    _list_workspaces.clear()


def _list_offsets(offsets):
    """host int64 array of n_lists + 1 pair offsets (kept alive by the caller's frame during the call)"""
    import numpy as np

    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    if off.ndim != 1 or off.size < 1 or off[0] != 0 or (np.diff(off) < 0).any():
        raise ValueError("list offsets must start at 0 and be non-decreasing")
    return off


def _lists_ids(query, doc, store, pair_q, pair_d):
    """(q64, d64, q32, d32, pq, pd, B, Q, L, device): one of the two id layouts"""
    if store is not None:
        qt, dt, pq, pd = _i32(store.q_table), _i32(store.d_table), _i32(pair_q), _i32(pair_d)
        return None, None, qt, dt, pq, pd, pq.numel(), qt.shape[1], dt.shape[1], pq.device
    q, d = _i64(query), _i64(doc)
    return q, d, None, None, None, None, q.shape[0], q.shape[1], d.shape[1], q.device


def knrm_forward_lists(offsets, packed, V, D, mu, sigma, w1, b1, w2=None, b2=None, scoretanh=False, query=None, doc=None, store=None, pair_q=None,
                       pair_d=None, out=None, check=True):
    """KNRM over whole candidate lists (capamd_knrm_forward_lists): pairs laid out list after list, `offsets` their n_lists + 1
    boundaries on the host; ids as [B, Q] / [B, L] tensors or through a CandidateStore (store, pair_q, pair_d)."""
    q, d, qt, dt, pq, pd, B, Q, L, dev = _lists_ids(query, doc, store, pair_q, pair_d)
    _need_gpu(packed, mu, sigma, w1, b1, w2, b2)
    off = _list_offsets(offsets)
    if int(off[-1]) != B:
        raise ValueError("the last list offset must be the number of pairs")
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=dev)
    hidden = 0 if w2 is None else w1.shape[0]
    st, ws = status_word(dev), _lists_workspace(dev, off.size - 1, V, B, L, Q)
    rc = _lib.load().capamd_knrm_forward_lists(
        _ptr(q), _ptr(d), _ptr(qt), _ptr(dt), _ptr(pq), _ptr(pd), ctypes.c_void_p(off.ctypes.data), off.size - 1, Q, L, _ptr(packed), V, D, _ptr(mu),
        _ptr(sigma), mu.numel(), _ptr(w1), _ptr(b1), hidden, _ptr(w2), _ptr(b2), int(bool(scoretanh)), _ptr(out), _ptr(st.t), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "capamd_knrm_forward_lists")
    if check:
        st.raise_if_set()
    return out


def drmm_forward_lists(offsets, idf, packed, V, D, edges, hist_type, gate_type, gate_w, emb_raw, w1, b1, w2, b2, out_w, out_b, query=None, doc=None,
                       store=None, pair_q=None, pair_d=None, out=None, counts_out=None, check=True):
    """DRMM over whole candidate lists (capamd_drmm_forward_lists); `idf`: [B, Q] per pair, or the store's [NQ, Q] table in indexed mode."""
    q, d, qt, dt, pq, pd, B, Q, L, dev = _lists_ids(query, doc, store, pair_q, pair_d)
    _need_gpu(packed, edges, gate_w, w1, b1, w2, b2, out_w, out_b)
    off = _list_offsets(offsets)
    if int(off[-1]) != B:
        raise ValueError("the last list offset must be the number of pairs")
    idf = _f32(idf)
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=dev)
    ld = emb_raw.stride(0) if emb_raw is not None else 0
    st, ws = status_word(dev), _lists_workspace(dev, off.size - 1, V, B, L, Q)
    rc = _lib.load().capamd_drmm_forward_lists(
        _ptr(q), _ptr(d), _ptr(qt), _ptr(dt), _ptr(pq), _ptr(pd), _ptr(idf), ctypes.c_void_p(off.ctypes.data), off.size - 1, Q, L, _ptr(packed), V, D,
        _ptr(edges), edges.numel(), HIST_TYPES[hist_type], GATE_TYPES[gate_type], _ptr(gate_w), _ptr(emb_raw), ld, _ptr(w1), _ptr(b1), w1.shape[0],
        _ptr(w2), _ptr(b2), _ptr(out_w), _ptr(out_b), _ptr(out), _ptr(counts_out), _ptr(st.t), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "capamd_drmm_forward_lists")
    if check:
        st.raise_if_set()
    return out


class AdamStep:
    """What a fused training step needs of a torch.optim.Adam: the parameters' device pointers with their moments (created the way
    Adam's own first step creates them), the host-side step count, and the step's scalars computed in double as the optimizer's
    non-capturable path does (torch/optim/adam.py: bias_correction = 1 - beta ** step; step_size = lr / bias_correction1;
    denom = sqrt(v) / sqrt(bias_correction2) + eps).  The state stays a plain Adam state_dict: checkpoints written during fused training
    load into the reference's trainer and vice versa."""

    def __init__(self, optimizer, params):
        if not isinstance(optimizer, torch.optim.Adam) or len(optimizer.param_groups) != 1:
            raise NotImplementedError("fused training steps follow torch.optim.Adam with one parameter group")
        g = optimizer.param_groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad") or g.get("maximize") or g.get("capturable") or torch.is_tensor(g["lr"]):
            raise NotImplementedError("fused training steps follow the plain Adam of the reference (no weight decay / amsgrad / capturable)")
        self.optimizer, self.group, self.params = optimizer, g, list(params)
        in_opt = {id(p) for p in g["params"]}
        # every tensor the optimizer trains must be one the device step updates: a parameter added to a model later would otherwise
        # silently never move under the fused step - the trainer falls back to the captured-graph route on this error
        missing = in_opt - {id(p) for p in self.params if p is not None}
        if missing:
            raise NotImplementedError(f"{len(missing)} optimizer parameter(s) are not covered by this reranker's device training step")
        ptrs = [[], [], []]
        for p in self.params:
            if p is None or id(p) not in in_opt:        # not trained (requires_grad False): no moments
                for col, v in zip(ptrs, (p.data_ptr() if p is not None else 0, 0, 0)):
                    col.append(v)
                continue
            st = optimizer.state[p]
            if len(st) == 0:       # (Adam._init_group)
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if st["step"].is_cuda:
                raise NotImplementedError("fused training steps keep Adam's step count on the host")
            for col, v in zip(ptrs, (p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr())):
                col.append(v)
        dev = next(p for p in self.params if p is not None).device
        self.table = torch.tensor(ptrs[0] + ptrs[1] + ptrs[2], dtype=torch.int64).to(dev)
        self.key = tuple(ptrs[0] + ptrs[1] + ptrs[2])
        self.trained = [p for p in self.params if p is not None and id(p) in in_opt]

    def still_valid(self):
        st = self.optimizer.state
        return all(len(st[p]) and st[p]["exp_avg"].data_ptr() in self.key for p in self.trained)

    def advance(self):
        """(step_size, 1 - beta1, beta2, eps, sqrt(bias_correction2)) of the NEXT step.  The step counters move on in `commit()`, after
        the C call has validated its arguments and launched: an EngineError must not leave the count ahead of the moments."""
        b1, b2 = self.group["betas"]
        t = float(self.optimizer.state[self.trained[0]]["step"]) + 1.0
        return float(self.group["lr"]) / (1 - b1 ** t), 1 - b1, b2, float(self.group["eps"]), (1 - b2 ** t) ** 0.5

    def commit(self):
        for p in self.trained:
            self.optimizer.state[p]["step"] += 1


_step_workspaces = {}


def knrm_train_step(query, posdoc, negdoc, packed, V, D, K, adam, train_kernels, scoretanh, softmax, check=True):
    """capamd_knrm_train_step: score(pos), score(neg), the trainer's pairwise loss, backward, Adam - on the device; returns the loss [1]."""
    _need_gpu(query, posdoc, negdoc, packed)
    q, dp, dn = _i64(query), _i64(posdoc), _i64(negdoc)
    B, Q = q.shape
    L = dp.shape[1]
    lib = _lib.load()
    ws = _step_workspace(q.device, int(lib.capamd_knrm_train_step_workspace_floats(B, K)))
    loss = torch.empty(1, dtype=torch.float32, device=q.device)
    step_size, omb1, b2, eps, bc2s = adam.advance()
    st = status_word(q.device)
    rc = lib.capamd_knrm_train_step(_ptr(q), _ptr(dp), _ptr(dn), B, Q, L, _ptr(packed), V, D, K, _ptr(adam.table), int(bool(train_kernels)),
                                    int(bool(scoretanh)), int(bool(softmax)), step_size, omb1, b2, eps, bc2s, _ptr(loss), _ptr(ws), ws.numel(),
                                    _ptr(st.t), _stream())
    _lib.check(rc, "capamd_knrm_train_step")
    adam.commit()
    if check:
        st.raise_if_set()
    return loss


def convknrm_train_step(query, posdoc, negdoc, emb, G, F, K, crossmatch, adam, scoretanh, softmax, check=True):
    """capamd_convknrm_train_step: score(pos), score(neg), the trainer's pairwise loss, backward, Adam - on the device; returns the loss [1]."""
    _need_gpu(query, posdoc, negdoc, emb)
    q, dp, dn, e = _i64(query), _i64(posdoc), _i64(negdoc), _f32(emb.detach())
    B, Q = q.shape
    L = dp.shape[1]
    q2, d2 = torch.cat([q, q]), torch.cat([dp, dn])
    lib = _lib.load()
    ws = _step_workspace(q.device, int(lib.capamd_convknrm_train_step_workspace_floats(B, Q, L, e.shape[1], G, F, K, int(bool(crossmatch)))))
    loss = torch.empty(1, dtype=torch.float32, device=q.device)
    step_size, omb1, b2, eps, bc2s = adam.advance()
    ptrs = (ctypes.c_void_p * len(adam.key))(*[p or None for p in adam.key])
    st = status_word(q.device)
    rc = lib.capamd_convknrm_train_step(_ptr(q2), _ptr(d2), B, Q, L, _ptr(e), e.shape[0], e.shape[1], G, F, K, int(bool(crossmatch)), ptrs, int(bool(scoretanh)),
                                        int(bool(softmax)), step_size, omb1, b2, eps, bc2s, _ptr(loss), _ptr(ws), ws.numel(), _ptr(st.t), _stream())
    _lib.check(rc, "capamd_convknrm_train_step")
    adam.commit()
    if check:
        st.raise_if_set()
    return loss


def pacrr_train_step_fits(B, Q, n_ngrams, kmax, use_idf, combine):
    """whether a batch's activations fit the one-workgroup stage of capamd_pacrr_train_step (include/capreolus_amd.h)"""
    H, X = int(combine), int(Q) * (int(n_ngrams) * int(kmax) + int(bool(use_idf)))
    return H <= 128 and H * X + H * H + H + 2 * B * X + 8 * B * H + 5 * B <= 36 * 1024


def pacrr_train_step(query, posdoc, negdoc, idf, packed, V, D, mingram, maxgram, nfilters, kmax, use_idf, combine, nonlinearity, adam, softmax, check=True):
    """capamd_pacrr_train_step: score(pos), score(neg), the trainer's pairwise loss, backward, Adam - on the device; returns the loss [1]."""
    _need_gpu(query, posdoc, negdoc, idf, packed)
    q, dp, dn, idf = _i64(query), _i64(posdoc), _i64(negdoc), _f32(idf)
    B, Q = q.shape
    L = dp.shape[1]
    q2, d2, idf2 = torch.cat([q, q]), torch.cat([dp, dn]), torch.cat([idf, idf])
    lib = _lib.load()
    ws = _step_workspace(q.device, int(lib.capamd_pacrr_train_step_workspace_floats(B, Q, L, int(mingram), int(maxgram), int(nfilters), int(kmax))))
    loss = torch.empty(1, dtype=torch.float32, device=q.device)
    step_size, omb1, b2, eps, bc2s = adam.advance()
    ptrs = (ctypes.c_void_p * len(adam.key))(*[p or None for p in adam.key])
    st = status_word(q.device)
    rc = lib.capamd_pacrr_train_step(_ptr(q2), _ptr(d2), _ptr(idf2), B, Q, L, _ptr(packed), V, D, int(mingram), int(maxgram), int(nfilters), int(kmax),
                                     int(bool(use_idf)), int(combine), NONLINEARITIES[nonlinearity], ptrs, int(bool(softmax)), step_size, omb1, b2, eps, bc2s,
                                     _ptr(loss), _ptr(ws), ws.numel(), _ptr(st.t), _stream())
    _lib.check(rc, "capamd_pacrr_train_step")
    adam.commit()
    if check:
        st.raise_if_set()
    return loss


def _step_workspace(device, n):
    key = (device.index, int(torch.cuda.current_stream(device).cuda_stream))
    ws = _step_workspaces.get(key)
    if ws is None or ws.numel() < n:
        ws = _step_workspaces[key] = torch.empty(n, dtype=torch.float32, device=device)
    return ws


def drmm_train_step(query, posdoc, negdoc, idf, packed, V, D, edges, hist_type, nodes, adam, softmax, check=True):
    """capamd_drmm_train_step: score(pos), score(neg), the trainer's pairwise loss, backward, Adam - on the device; returns the loss [1]."""
    _need_gpu(query, posdoc, negdoc, idf, packed, edges)
    q, dp, dn, idf = _i64(query), _i64(posdoc), _i64(negdoc), _f32(idf)
    B, Q = q.shape
    L = dp.shape[1]
    lib = _lib.load()
    ws = _step_workspace(q.device, int(lib.capamd_drmm_train_step_workspace_floats(B, Q, edges.numel(), int(nodes))))
    loss = torch.empty(1, dtype=torch.float32, device=q.device)
    step_size, omb1, b2, eps, bc2s = adam.advance()
    st = status_word(q.device)
    rc = lib.capamd_drmm_train_step(_ptr(q), _ptr(dp), _ptr(dn), _ptr(idf), B, Q, L, _ptr(packed), V, D, _ptr(edges), edges.numel(), HIST_TYPES[hist_type],
                                    int(nodes), _ptr(adam.table), int(bool(softmax)), step_size, omb1, b2, eps, bc2s, _ptr(loss), _ptr(ws), ws.numel(),
                                    _ptr(st.t), _stream())
    _lib.check(rc, "capamd_drmm_train_step")
    adam.commit()
    if check:
        st.raise_if_set()
    return loss


def drmmtks_train_step(query, posdoc, negdoc, idf, packed, V, D, topk, adam, softmax, check=True):
    """capamd_drmmtks_train_step: score(pos), score(neg), the trainer's pairwise loss, backward, Adam - on the device; returns the loss [1]."""
    _need_gpu(query, posdoc, negdoc, idf, packed)
    q, dp, dn, idf = _i64(query), _i64(posdoc), _i64(negdoc), _f32(idf)
    B, Q = q.shape
    L = dp.shape[1]
    lib = _lib.load()
    ws = _step_workspace(q.device, int(lib.capamd_drmmtks_train_step_workspace_floats(B, Q, int(topk))))
    loss = torch.empty(1, dtype=torch.float32, device=q.device)
    step_size, omb1, b2, eps, bc2s = adam.advance()
    st = status_word(q.device)
    rc = lib.capamd_drmmtks_train_step(_ptr(q), _ptr(dp), _ptr(dn), _ptr(idf), B, Q, L, _ptr(packed), V, D, int(topk), _ptr(adam.table), int(bool(softmax)),
                                       step_size, omb1, b2, eps, bc2s, _ptr(loss), _ptr(ws), ws.numel(), _ptr(st.t), _stream())
    _lib.check(rc, "capamd_drmmtks_train_step")
    adam.commit()
    if check:
        st.raise_if_set()
    return loss


def knrm_features(query, doc, packed, V, D, mu, sigma, need_grad=True, check=True):
    """capamd_knrm_features: kernel-pooling features [B, K] and d f/d mu, d f/d sigma [B, K] (None when not needed)."""
    _need_gpu(query, doc, packed, mu, sigma)
    q, d = _i64(query), _i64(doc)
    B, Q = q.shape
    L, K = d.shape[1], mu.numel()
    feat = torch.empty((B, K), dtype=torch.float32, device=q.device)
    dmu = torch.empty_like(feat) if need_grad else None
    dsg = torch.empty_like(feat) if need_grad else None
    st = status_word(q.device)
    rc = _lib.load().capamd_knrm_features(_ptr(q), _ptr(d), B, Q, L, _ptr(packed), V, D, _ptr(mu), _ptr(sigma), K, _ptr(feat),
                                          _ptr(dmu), _ptr(dsg), _ptr(st.t), _stream())
    _lib.check(rc, "capamd_knrm_features")
    if check:
        st.raise_if_set()
    return feat, dmu, dsg


class KnrmFeatures(torch.autograd.Function):
    """features = f(query, doc; mu, sigma) with the HIP kernel supplying both the value and the two Jacobian diagonals
    (f_k depends on mu_k / sigma_k only), so `combine` and the loss can sit on top of it under autograd."""

    @staticmethod
    def forward(ctx, mu, sigma, query, doc, packed, V, D):
        need = mu.requires_grad or sigma.requires_grad
        feat, dmu, dsg = knrm_features(query, doc, packed, V, D, mu.detach().float().contiguous(), sigma.detach().float().contiguous(),
                                       need_grad=need)
        if need:
            ctx.save_for_backward(dmu, dsg)
        ctx.has = need
        return feat

    @staticmethod
    def backward(ctx, g):
        if not ctx.has:
            return (None,) * 7
        dmu, dsg = ctx.saved_tensors
        return (g * dmu).sum(0), (g * dsg).sum(0), None, None, None, None, None


def similarity_matrix_autograd(embedding, query, doc):
    """SimilarityMatrix.forward (capreolus/reranker/common.py:155-182) as ATen ops under autograd ON THE GPU, for the configurations in
    which the embedding table trains (KNRM finetune=True, DRMM-TKS freezeemb=False): the gradient has to reach the [V, D] table through
    both operands of every cosine, which the HIP front end (frozen tables only) does not provide.  Exact matches of OOV terms (negative
    ids) + cosine of in-vocabulary terms (everything else looks up row 0), pads zeroed.  -> [B, Q, L]."""
    _need_gpu(query, doc, embedding.weight)
    q, d = query.long(), doc.long()
    qo, do = q.clamp(max=0), d.clamp(max=0)
    exact = (qo[:, :, None] == do[:, None, :]).float().masked_fill((qo == 0)[:, :, None], 0.0).masked_fill((do == 0)[:, None, :], 0.0)
    qi, di = q.clamp(min=0), d.clamp(min=0)
    a, b = embedding(qi), embedding(di)
    den = (a.norm(p=2, dim=2)[:, :, None] + 1e-9) * (b.norm(p=2, dim=2)[:, None, :] + 1e-9)
    cos = (a.bmm(b.permute(0, 2, 1)) / den).masked_fill((qi == 0)[:, :, None], 0.0).masked_fill((di == 0)[:, None, :], 0.0)
    return exact + cos


class NgramConv(torch.autograd.Function):
    """ConvKNRM's n-gram convolutions over the frozen embedding table, differentiable in their weights and biases
    (capamd_ngram_conv_forward / _backward; ConvKNRM.py:42-51 under the trainer's loss.backward()).
    apply(q_ids [N, Q], d_ids [N, L], emb [V, D], w_1, b_1, ..., w_G, b_G) -> qrep [N, G, Q, F], drep [N, G, L, F]."""

    @staticmethod
    def forward(ctx, q_ids, d_ids, emb, *wb):
        _need_gpu(q_ids, d_ids, emb, *wb)
        qi, di, e = _i64(q_ids), _i64(d_ids), _f32(emb.detach())
        ws, bs = [_f32(w.detach()) for w in wb[0::2]], [_f32(b.detach()) for b in wb[1::2]]
        G, F, D = len(ws), ws[0].shape[0], e.shape[1]
        N, Q, L = qi.shape[0], qi.shape[1], di.shape[1]
        lib = _lib.load()
        qrep = torch.empty((N, G, Q, F), dtype=torch.float32, device=e.device)
        drep = torch.empty((N, G, L, F), dtype=torch.float32, device=e.device)
        work = _step_workspace(e.device, int(lib.capamd_ngram_conv_workspace_floats(D, G, F, 0)))
        wp, bp = (ctypes.c_void_p * G)(*[w.data_ptr() for w in ws]), (ctypes.c_void_p * G)(*[b.data_ptr() for b in bs])
        st = status_word(e.device)
        rc = lib.capamd_ngram_conv_forward(_ptr(qi), _ptr(di), N, Q, L, _ptr(e), e.shape[0], D, wp, bp, G, F, _ptr(qrep), _ptr(drep), _ptr(work),
                                           work.numel(), _ptr(st.t), _stream())
        _lib.check(rc, "capamd_ngram_conv_forward")
        st.raise_if_set()
        ctx.save_for_backward(qi, di, e)
        ctx.geom = (G, F, [tuple(w.shape) for w in ws])
        return qrep, drep

    @staticmethod
    def backward(ctx, dq, dd):
        qi, di, e = ctx.saved_tensors
        G, F, shapes = ctx.geom
        N, Q, L, D = qi.shape[0], qi.shape[1], di.shape[1], e.shape[1]
        lib = _lib.load()
        dws = [torch.empty(sh, dtype=torch.float32, device=e.device) for sh in shapes]
        dbs = [torch.empty(F, dtype=torch.float32, device=e.device) for _ in shapes]
        work = _step_workspace(e.device, int(lib.capamd_ngram_conv_workspace_floats(D, G, F, 1)))
        wp, bp = (ctypes.c_void_p * G)(*[w.data_ptr() for w in dws]), (ctypes.c_void_p * G)(*[b.data_ptr() for b in dbs])
        dq, dd = _f32(dq), _f32(dd)
        st = status_word(e.device)
        rc = lib.capamd_ngram_conv_backward(_ptr(qi), _ptr(di), N, Q, L, _ptr(e), e.shape[0], D, G, F, _ptr(dq), _ptr(dd), wp, bp, _ptr(work),
                                            work.numel(), _ptr(st.t), _stream())
        _lib.check(rc, "capamd_ngram_conv_backward")
        out = [None, None, None]
        for w, b in zip(dws, dbs):
            out += [w, b]
        return tuple(out)


class KernelPool(torch.autograd.Function):
    """Cosine similarity + RBF kernel pooling over dense n-gram views, differentiable in both operands and in (mu, sigma)
    (capamd_kernel_pool_forward / _backward): ConvKNRM's training step between its convolutions and `combine`.
    qrep [B, GQ, Q, F], drep [B, GD, L, F] -> features [B, K V] in the reference's order (ConvKNRM.py:66-75)."""

    @staticmethod
    def forward(ctx, qrep, drep, q_ids, d_ids, mu, sigma, crossmatch):
        _need_gpu(qrep, drep, q_ids, d_ids, mu, sigma)
        qr, dr, qi, di = _f32(qrep.detach()), _f32(drep.detach()), _i64(q_ids), _i64(d_ids)
        m, sg = _f32(mu.detach()), _f32(sigma.detach())
        B, GQ, Q, F = qr.shape
        GD, L = dr.shape[1], dr.shape[2]
        K = m.numel()
        V = GQ * GD if crossmatch else GD
        T = (GQ if crossmatch else 1) * Q
        feat = torch.empty((B, K * V), dtype=torch.float32, device=qr.device)
        ksum = torch.empty((B, GD, T, K), dtype=torch.float32, device=qr.device)
        rowsum = torch.empty((B, GD, T), dtype=torch.float32, device=qr.device)
        C = int(_lib.load().capamd_kernel_pool_chunks(L))
        sums = torch.empty((B, GD, C, T, K + 1), dtype=torch.float32, device=qr.device)
        rc = _lib.load().capamd_kernel_pool_forward(_ptr(qr), _ptr(dr), _ptr(qi), _ptr(di), B, GQ, GD, Q, L, F, int(bool(crossmatch)), _ptr(m), _ptr(sg), K,
                                                    _ptr(feat), _ptr(ksum), _ptr(rowsum), _ptr(sums), _stream())
        _lib.check(rc, "capamd_kernel_pool_forward")
        ctx.save_for_backward(qr, dr, qi, di, m, sg, ksum, rowsum)
        ctx.crossmatch = bool(crossmatch)
        return feat

    @staticmethod
    def backward(ctx, g):
        qr, dr, qi, di, m, sg, ksum, rowsum = ctx.saved_tensors
        B, GQ, Q, F = qr.shape
        GD, L = dr.shape[1], dr.shape[2]
        K, T = m.numel(), ksum.shape[2]
        C = int(_lib.load().capamd_kernel_pool_chunks(L))
        dq_part = torch.empty((B, GD, C, T, F), dtype=torch.float32, device=qr.device)
        dd = torch.empty_like(dr)
        dmu = torch.empty((B * GD * C, K), dtype=torch.float32, device=qr.device)
        dsg = torch.empty_like(dmu)
        rc = _lib.load().capamd_kernel_pool_backward(_ptr(qr), _ptr(dr), _ptr(qi), _ptr(di), B, GQ, GD, Q, L, F, int(ctx.crossmatch), _ptr(m), _ptr(sg), K,
                                                     _ptr(_f32(g)), _ptr(ksum), _ptr(rowsum), _ptr(dq_part), _ptr(dd), _ptr(dmu), _ptr(dsg), _stream())
        _lib.check(rc, "capamd_kernel_pool_backward")
        # (without crossmatch block gd holds query view gd: T = Q)
        dq = dq_part.view(B, GD * C, GQ, Q, F).sum(1) if ctx.crossmatch else dq_part.sum(2)
        return dq, dd, None, None, dmu.sum(0), dsg.sum(0), None


class PacrrConvMax(torch.autograd.Function):
    """PACRR's n-gram Conv2d -> ReLU -> max over filters -> k-max over the document on a [B, Q, L] similarity matrix, with the
    gradient into the convolution weights (capamd_pacrr_convmax_forward / _backward; PACRR.py:68-78).  No gradient into `sim`:
    the embedding table behind it is frozen (PACRR.py:26)."""

    @staticmethod
    def forward(ctx, sim, conv_w, conv_b, mingram, maxgram, nfilters, kmax):
        _need_gpu(sim, conv_w, conv_b)
        x, w, b = _f32(sim.detach()), _f32(conv_w.detach()), _f32(conv_b.detach())
        B, Q, L = x.shape
        n = (maxgram - mingram + 1) * kmax
        top = torch.empty((B, Q, n), dtype=torch.float32, device=x.device)
        pos = torch.empty((B, Q, n), dtype=torch.int32, device=x.device)
        filt = torch.empty_like(pos)
        rc = _lib.load().capamd_pacrr_convmax_forward(_ptr(x), B, Q, L, mingram, maxgram, nfilters, kmax, _ptr(w), _ptr(b), _ptr(top), _ptr(pos), _ptr(filt),
                                                      _stream())
        _lib.check(rc, "capamd_pacrr_convmax_forward")
        ctx.save_for_backward(x, pos, filt)
        ctx.geom = (mingram, maxgram, nfilters, kmax, w.numel(), b.numel())
        return top

    @staticmethod
    def backward(ctx, g):
        x, pos, filt = ctx.saved_tensors
        mingram, maxgram, nfilters, kmax, nw, nb = ctx.geom
        B, Q, L = x.shape
        dw = torch.empty(nw, dtype=torch.float32, device=x.device)
        db = torch.empty(nb, dtype=torch.float32, device=x.device)
        rc = _lib.load().capamd_pacrr_convmax_backward(_ptr(x), B, Q, L, mingram, maxgram, nfilters, kmax, _ptr(_f32(g)), _ptr(pos), _ptr(filt), _ptr(dw),
                                                       _ptr(db), _stream())
        _lib.check(rc, "capamd_pacrr_convmax_backward")
        return None, dw, db, None, None, None, None


def drmm_features(query, doc, packed, V, D, edges, hist_type, check=True):
    """capamd_drmm_features: matching-histogram features [B, Q, nbins+1] (no trainable inputs)."""
    _need_gpu(query, doc, packed, edges)
    q, d = _i64(query), _i64(doc)
    B, Q = q.shape
    feat = torch.empty((B, Q, edges.numel() + 1), dtype=torch.float32, device=q.device)
    st = status_word(q.device)
    rc = _lib.load().capamd_drmm_features(_ptr(q), _ptr(d), B, Q, d.shape[1], _ptr(packed), V, D, _ptr(edges), edges.numel(),
                                          HIST_TYPES[hist_type], _ptr(feat), _ptr(st.t), _stream())
    _lib.check(rc, "capamd_drmm_features")
    if check:
        st.raise_if_set()
    return feat


def drmmtks_forward(query, doc, idf, packed, V, D, topk, gate_w, ffw_w, ffw_b, out_w, out_b, out=None, check=True):
    """DRMMTKS_class.forward (reference DRMMTKS.py:50-64): fp32 [B]."""
    _need_gpu(query, doc, idf, packed, gate_w, ffw_w, ffw_b, out_w, out_b)
    q, d, idf = _i64(query), _i64(doc), _f32(idf)
    B, Q = q.shape
    L = d.shape[1]
    if topk > L:
        raise RuntimeError("selected index k out of range")  # what torch.topk raises at DRMMTKS.py:56
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=q.device)
    st = status_word(q.device)
    rc = _lib.load().capamd_drmmtks_forward(_ptr(q), _ptr(d), _ptr(idf), B, Q, L, _ptr(packed), V, D, int(topk), _ptr(gate_w), _ptr(ffw_w),
                                            _ptr(ffw_b), _ptr(out_w), _ptr(out_b), _ptr(out), _ptr(st.t), _stream())
    _lib.check(rc, "capamd_drmmtks_forward")
    if check:
        st.raise_if_set()
    return out


def drmmtks_forward_lists(offsets, idf, packed, V, D, topk, gate_w, ffw_w, ffw_b, out_w, out_b, query=None, doc=None, store=None, pair_q=None,
                          pair_d=None, out=None, check=True):
    """DRMM-TKS over whole candidate lists (capamd_drmmtks_forward_lists); `idf`: [B, Q] per pair, or the store's [NQ, Q] table."""
    q, d, qt, dt, pq, pd, B, Q, L, dev = _lists_ids(query, doc, store, pair_q, pair_d)
    _need_gpu(packed, gate_w, ffw_w, ffw_b, out_w, out_b)
    off = _list_offsets(offsets)
    if int(off[-1]) != B:
        raise ValueError("the last list offset must be the number of pairs")
    if topk > L:
        raise RuntimeError("selected index k out of range")  # what torch.topk raises at DRMMTKS.py:56
    idf = _f32(idf)
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=dev)
    st, ws = status_word(dev), _lists_workspace(dev, off.size - 1, V, B, L, Q)
    rc = _lib.load().capamd_drmmtks_forward_lists(
        _ptr(q), _ptr(d), _ptr(qt), _ptr(dt), _ptr(pq), _ptr(pd), _ptr(idf), ctypes.c_void_p(off.ctypes.data), off.size - 1, Q, L, _ptr(packed), V, D,
        int(topk), _ptr(gate_w), _ptr(ffw_w), _ptr(ffw_b), _ptr(out_w), _ptr(out_b), _ptr(out), _ptr(st.t), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "capamd_drmmtks_forward_lists")
    if check:
        st.raise_if_set()
    return out


def drmmtks_features(query, doc, packed, V, D, topk, check=True):
    """The sorted top-k similarities of every query term (DRMMTKS.py:55-56): fp32 [B, Q, topk] (capamd_drmmtks_features)."""
    _need_gpu(query, doc, packed)
    q, d = _i64(query), _i64(doc)
    B, Q = q.shape
    L = d.shape[1]
    if topk > L:
        raise RuntimeError("selected index k out of range")
    out = torch.empty((B, Q, topk), dtype=torch.float32, device=q.device)
    st = status_word(q.device)
    rc = _lib.load().capamd_drmmtks_features(_ptr(q), _ptr(d), B, Q, L, _ptr(packed), V, D, int(topk), _ptr(out), _ptr(st.t), _stream())
    _lib.check(rc, "capamd_drmmtks_features")
    if check:
        st.raise_if_set()
    return out


NONLINEARITIES = {"none": 0, "relu": 1, "tanh": 2}


def pacrr_forward(query, doc, idf, packed, V, D, mingram, maxgram, nfilters, kmax, conv_w, conv_b, use_idf, nonlinearity, w1, b1, w2, b2,
                  w3, b3, out=None, check=True):
    """PACRR_class.forward (reference PACRR.py:42-55): fp32 [B].  conv_w / conv_b: the n-gram modules' Conv2d weights /
    biases, flattened back to back (capamd_pacrr_forward)."""
    _need_gpu(query, doc, idf, packed, conv_w, conv_b, w1, b1, w2, b2, w3, b3)
    q, d, idf = _i64(query), _i64(doc), _f32(idf)
    B, Q = q.shape
    L = d.shape[1]
    if kmax > L:
        raise RuntimeError("selected index k out of range")  # what torch.topk raises at PACRR.py:74
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=q.device)
    st = status_word(q.device)
    rc = _lib.load().capamd_pacrr_forward(_ptr(q), _ptr(d), _ptr(idf), B, Q, L, _ptr(packed), V, D, int(mingram), int(maxgram), int(nfilters),
                                          int(kmax), _ptr(conv_w), _ptr(conv_b), int(bool(use_idf)), w1.shape[0], NONLINEARITIES[nonlinearity],
                                          _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), _ptr(out), _ptr(st.t), _stream())
    _lib.check(rc, "capamd_pacrr_forward")
    if check:
        st.raise_if_set()
    return out


def pacrr_forward_lists(offsets, idf, packed, V, D, mingram, maxgram, nfilters, kmax, conv_w, conv_b, use_idf, nonlinearity, w1, b1, w2, b2, w3, b3,
                        query=None, doc=None, store=None, pair_q=None, pair_d=None, out=None, check=True, pair_part=True):
    """PACRR over whole candidate lists (capamd_pacrr_forward_lists); `idf`: [B, Q] per pair, or the store's [NQ, Q] table.
    `pair_part=False` hands the library a workspace without the per-pair part (capamd_lists_workspace_bytes with n_pairs = 0): the combine
    layers then run at the end of every pair's convolution workgroup instead of in one pass behind them - the same scores, bit for bit."""
    q, d, qt, dt, pq, pd, B, Q, L, dev = _lists_ids(query, doc, store, pair_q, pair_d)
    _need_gpu(packed, conv_w, conv_b, w1, b1, w2, b2, w3, b3)
    off = _list_offsets(offsets)
    if int(off[-1]) != B:
        raise ValueError("the last list offset must be the number of pairs")
    if kmax > L:
        raise RuntimeError("selected index k out of range")  # what torch.topk raises at PACRR.py:74
    idf = _f32(idf)
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=dev)
    st, ws = status_word(dev), _lists_workspace(dev, off.size - 1, V, B if pair_part else 0, L)
    rc = _lib.load().capamd_pacrr_forward_lists(
        _ptr(q), _ptr(d), _ptr(qt), _ptr(dt), _ptr(pq), _ptr(pd), _ptr(idf), ctypes.c_void_p(off.ctypes.data), off.size - 1, Q, L, _ptr(packed), V, D,
        int(mingram), int(maxgram), int(nfilters), int(kmax), _ptr(conv_w), _ptr(conv_b), int(bool(use_idf)), w1.shape[0], NONLINEARITIES[nonlinearity],
        _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), _ptr(out), _ptr(st.t), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "capamd_pacrr_forward_lists")
    if check:
        st.raise_if_set()
    return out


class ConvProjectionTables:
    """ConvKNRM's Conv1d layers folded into per-token projection tables (capamd_convknrm_pack_tables).  Rebuilt whenever the
    embedding table or any convolution parameter changes (version counters + storage pointers)."""

    def __init__(self):
        self._key = None
        self.tables = None

    def get(self, weight, conv_ws, conv_bs):
        _need_gpu(weight, *conv_ws, *conv_bs)
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (weight, *conv_ws, *conv_bs)) + (weight.device.index,)
        if key != self._key:
            lib = _lib.load()
            w = _f32(weight.detach())
            V, D = w.shape
            G, F = len(conv_ws), conv_ws[0].shape[0]
            nbytes = lib.capamd_convknrm_table_bytes(V, G, F)
            if nbytes < 0:
                raise ValueError(f"ConvKNRM geometry maxngram={G}, filters={F} is not supported (maxngram <= 3, filters a multiple of 16 up to 128)")
            cw = torch.cat([_f32(c.detach()).reshape(-1) for c in conv_ws]).contiguous()
            cb = torch.cat([_f32(c.detach()).reshape(-1) for c in conv_bs]).contiguous()
            tables = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
            _lib.check(lib.capamd_convknrm_pack_tables(_ptr(w), V, D, w.stride(0), _ptr(cw), _ptr(cb), G, F, _ptr(tables), _stream()),
                       "capamd_convknrm_pack_tables")
            self.tables, self._key = tables, key
        return self.tables


def convknrm_forward(query, doc, tables, V, maxngram, filters, crossmatch, mu, sigma, w1, b1, w2=None, b2=None, score_tanh=False, out=None,
                     check=True):
    """ConvKNRM_class.forward (reference ConvKNRM.py:42-77): fp32 [B].  w2 is None: single combine layer."""
    _need_gpu(query, doc, tables, mu, sigma, w1, b1)
    q, d = _i64(query), _i64(doc)
    B, Q = q.shape
    L = d.shape[1]
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=q.device)
    st = status_word(q.device)
    H = 0 if w2 is None else w1.shape[0]
    rc = _lib.load().capamd_convknrm_forward(_ptr(q), _ptr(d), B, Q, L, _ptr(tables), V, int(maxngram), int(filters), int(bool(crossmatch)),
                                             _ptr(mu), _ptr(sigma), mu.numel(), _ptr(w1), _ptr(b1), H, None if w2 is None else _ptr(w2),
                                             None if b2 is None else _ptr(b2), int(bool(score_tanh)), _ptr(out), _ptr(st.t), _stream())
    _lib.check(rc, "capamd_convknrm_forward")
    if check:
        st.raise_if_set()
    return out


def convknrm_forward_lists(offsets, query, doc, tables, V, maxngram, filters, crossmatch, mu, sigma, w1, b1, w2=None, b2=None, score_tanh=False, out=None,
                           check=True):
    """ConvKNRM over whole candidate lists (capamd_convknrm_forward_lists): pairs laid out list after list, `offsets` their n_lists + 1
    boundaries on the host; the unigram document view once per distinct token of a list.  Scores equal convknrm_forward's bit for bit."""
    _need_gpu(query, doc, tables, mu, sigma, w1, b1)
    q, d = _i64(query), _i64(doc)
    B, Q = q.shape
    L = d.shape[1]
    off = _list_offsets(offsets)
    if int(off[-1]) != B:
        raise ValueError("the last list offset must be the number of pairs")
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=q.device)
    lib = _lib.load()
    n_lists = off.size - 1
    one = int(lib.capamd_convknrm_lists_workspace_bytes(1, int(V), Q, int(maxngram), int(filters)))
    if one == 0:
        raise ValueError("capamd_convknrm_forward_lists does not take this geometry")
    per_list = int(lib.capamd_convknrm_lists_workspace_bytes(2, int(V), Q, int(maxngram), int(filters))) - one
    in_flight = max(1, min(n_lists, LISTS_WORKSPACE_BUDGET // max(per_list, 1)))
    nbytes = int(lib.capamd_convknrm_lists_workspace_bytes(in_flight, int(V), Q, int(maxngram), int(filters)))
    key = (q.device.index, int(torch.cuda.current_stream(q.device).cuda_stream))
    ws = _list_workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _list_workspaces[key] = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
    st = status_word(q.device)
    H = 0 if w2 is None else w1.shape[0]
    rc = lib.capamd_convknrm_forward_lists(_ptr(q), _ptr(d), ctypes.c_void_p(off.ctypes.data), n_lists, Q, L, _ptr(tables), V, int(maxngram), int(filters),
                                           int(bool(crossmatch)), _ptr(mu), _ptr(sigma), mu.numel(), _ptr(w1), _ptr(b1), H,
                                           None if w2 is None else _ptr(w2), None if b2 is None else _ptr(b2), int(bool(score_tanh)), _ptr(out),
                                           _ptr(st.t), _ptr(ws), nbytes, _stream())
    _lib.check(rc, "capamd_convknrm_forward_lists")
    if check:
        st.raise_if_set()
    return out


CLS_MODES = {None: 0, "avg": 1, "max": 2}


class CedrEngine:
    """CEDR-KNRM on top of a BertEngine's packed encoder: capamd_cedr_passage_features (encoder + per-layer masked cosine matrices +
    kernel pooling) followed by capamd_cedr_score (document-level log / sums, [CLS] feature, combine layers)."""

    def __init__(self, bert_engine):
        self.be = bert_engine
        self._wss = {}
        self._streams = {}

    def _passages(self, ids, mask, seg, S, qm0, maxqlen, layers, mu, sigma, pk, cls, ws_key):
        """One capamd_cedr_passage_features call over n passages of length S ([n, S] x3; qm0 [n, A]) on the current stream."""
        m, lib = self.be.model(), _lib.load()
        n = ids.shape[0]
        mb = min(self.be.microbatch * max(1, 256 // S), n)
        need = lib.capamd_bert_workspace_bytes(ctypes.byref(m), S, mb, n)
        if need < 0:
            raise ValueError(f"unsupported passage length {S} (supported: multiples of 32 up to 256, 384, 512)")
        ws = self._wss.get(ws_key)
        if ws is None or ws.numel() < need or ws.device != ids.device:
            ws = self._wss[ws_key] = torch.empty(need, dtype=torch.uint8, device=ids.device)
        arr = (ctypes.c_int * max(1, len(layers)))(*layers)
        rc = lib.capamd_cedr_passage_features(_ptr(ids), _ptr(mask), _ptr(seg), n, 1, S, ctypes.byref(m), mb, _ptr(ws), ws.numel(), int(maxqlen),
                                              _ptr(qm0), arr, len(layers), _ptr(mu), _ptr(sigma), mu.numel(), _ptr(pk), _ptr(cls),
                                              _ptr(status_word(ids.device).t), _stream())
        _lib.check(rc, "capamd_cedr_passage_features")

    def forward(self, doc_input, doc_mask, doc_seg, maxqlen, simmat_layers, mu, sigma, cls_mode, w1, b1, w2=None, b2=None, check=True,
                return_features=False, skip_padding=None):
        """CEDRKNRM_Class.forward (reference CEDRKNRM.py:151-185): int64 [B, P, S] x3 -> fp32 [B].

        skip_padding (default: the encoder engine's setting): as in BertEngine.forward, every passage is encoded at the shortest
        supported length that holds its last attended token.  Padded key positions get an attention weight of exactly 0 and
        padded document columns are masked out of every kernel sum, so the per-passage sums and [CLS] rows are those of the
        full-length computation; only the dead rows and columns are not computed."""
        _need_gpu(doc_input, doc_mask, doc_seg, mu, sigma, w1, b1)
        if cls_mode not in CLS_MODES:
            raise ValueError("cls must be 'avg', 'max' or None")
        ids, mask, seg = _i64(doc_input), _i64(doc_mask), _i64(doc_seg)
        B, P, S = ids.shape
        dev = ids.device
        out = torch.empty(B, dtype=torch.float32, device=dev)
        if B == 0:
            return out
        m = self.be.model()
        lib = _lib.load()
        # in the configured order: the reference concatenates the per-layer features as listed (CEDRKNRM.py:166-168), so the
        # columns of `combine` follow it; -1 alone = no similarity matrices (CEDRKNRM.py:48-52)
        layers = [int(x) for x in simmat_layers if int(x) >= 0]
        n_sel, K, A, H, NP = len(layers), mu.numel(), maxqlen + 1, m.hidden, B * P
        if A + 1 > S:
            raise ValueError("maxqlen + 2 exceeds the passage length")
        # the query mask the reference applies to every passage of a document: its FIRST passage's (CEDRKNRM.py:123)
        qm0 = ((mask[:, 0, 1:A + 1] != 0) & (seg[:, 0, 1:A + 1] == 0)).float().repeat_interleave(P, dim=0).contiguous()   # [NP, A]
        pk = torch.empty((max(1, n_sel), NP, K * A), dtype=torch.float32, device=dev)
        cls = torch.empty((NP, H), dtype=torch.float32, device=dev)
        fids, fmask, fseg = ids.view(NP, S), mask.view(NP, S), seg.view(NP, S)
        if skip_padding is None:
            skip_padding = self.be.skip_padding
        lengths = [x for x in SUPPORTED_LENGTHS if x < S and x > A + 1] if (skip_padding and S in SUPPORTED_LENGTHS) else []
        if not lengths:
            self._passages(fids, fmask, fseg, S, qm0, maxqlen, layers, mu, sigma, pk, cls, None)
        else:
            end = ((fmask != 0) * torch.arange(1, S + 1, device=dev)).amax(dim=1)
            bounds = lengths + [S]
            bucket = torch.bucketize(end, torch.tensor(lengths, device=dev), right=False)
            order = torch.argsort(bucket, stable=True)
            counts = torch.bincount(bucket, minlength=len(bounds)).cpu().tolist()   # the one host round trip of this call
            main = torch.cuda.current_stream(dev)
            lo, used = 0, []
            for Sb, n in zip(bounds, counts):
                if n == 0:
                    continue
                sel = order[lo:lo + n]
                lo += n
                side = self._streams.get(Sb)
                if side is None:
                    side = self._streams[Sb] = torch.cuda.Stream(device=dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    # (a ragged bucket is padded to whole GEMM tiles inside capamd_cedr_passage_features)
                    bi, bm, bs = (t.index_select(0, sel)[:, :Sb].contiguous() for t in (fids, fmask, fseg))
                    bq = qm0.index_select(0, sel)
                    pk_b = torch.empty((max(1, n_sel), n, K * A), dtype=torch.float32, device=dev)
                    cls_b = torch.empty((n, H), dtype=torch.float32, device=dev)
                    self._passages(bi, bm, bs, Sb, bq, maxqlen, layers, mu, sigma, pk_b, cls_b, Sb)
                    pk.index_copy_(1, sel, pk_b)
                    cls.index_copy_(0, sel, cls_b)
                    for t in (bi, bm, bs, bq, pk_b, cls_b, sel):
                        t.record_stream(side)
                used.append(side)
            for side in used:
                main.wait_stream(side)
        n_in = (H if cls_mode else 0) + n_sel * K
        feats = torch.empty((B, n_in), dtype=torch.float32, device=dev) if return_features else None
        hidden = 0 if w2 is None else w1.shape[0]
        rc = lib.capamd_cedr_score(_ptr(pk), _ptr(cls), B, P, int(maxqlen), n_sel, K, H, CLS_MODES[cls_mode], _ptr(w1), _ptr(b1), hidden,
                                   None if w2 is None else _ptr(w2), None if b2 is None else _ptr(b2), _ptr(out),
                                   None if feats is None else _ptr(feats), _stream())
        _lib.check(rc, "capamd_cedr_score")
        if check:
            status_word(dev).raise_if_set()
        return (out, feats) if return_features else out
