"""ctypes front end of oracle/liboracle.so (the C restatement in interaction_oracle.c).

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.  All arrays are host numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

HIST_TYPES = {"CH": 0, "NH": 1, "LCH": 2}
GATE_TYPES = {"IDF": 0, "TV": 1}


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "interaction_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.oracle_row_stride.restype = ctypes.c_int64
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def row_stride(D):
    return int(lib().oracle_row_stride(ctypes.c_int(D)))


def pack(emb):
    emb = _f32(emb)
    V, D = emb.shape
    packed = np.empty((V, row_stride(D)), dtype=np.float32)
    lib().oracle_pack(_p(emb), ctypes.c_int64(V), ctypes.c_int(D), ctypes.c_int64(D), _p(packed))
    return packed


def simmat(q_ids, d_ids, packed, D):
    q_ids, d_ids = _i64(q_ids), _i64(d_ids)
    B, Q = q_ids.shape
    L = d_ids.shape[1]
    out = np.empty((B, Q, L), dtype=np.float32)
    err = lib().oracle_simmat(_p(q_ids), _p(d_ids), B, Q, L, _p(packed), ctypes.c_int64(packed.shape[0]), D, _p(out))
    return out, err


def knrm(q_ids, d_ids, packed, D, mu, sigma, w1, b1, w2=None, b2=None, scoretanh=False):
    """w1 [1,K] & w2 None -> singlefc;  w1 [H,K], w2 [1,H] -> two layers (KNRM.py:27-34)."""
    q_ids, d_ids = _i64(q_ids), _i64(d_ids)
    B, Q = q_ids.shape
    L = d_ids.shape[1]
    mu, sigma, w1, b1 = _f32(mu), _f32(sigma), _f32(w1), _f32(b1)
    K = mu.shape[0]
    hidden = 0
    if w2 is not None:
        w2, b2 = _f32(w2), _f32(b2)
        hidden = w1.shape[0]
    out = np.empty(B, dtype=np.float32)
    err = lib().oracle_knrm(_p(q_ids), _p(d_ids), B, Q, L, _p(packed), ctypes.c_int64(packed.shape[0]), D, _p(mu),
                            _p(sigma), K, _p(w1), _p(b1), hidden, _p(w2), _p(b2), int(bool(scoretanh)), _p(out))
    return out, err


def drmm(q_ids, d_ids, idf, packed, D, edges, hist_type, gate_type, gate_w, emb_raw, w1, b1, w2, b2, out_w, out_b):
    q_ids, d_ids, idf = _i64(q_ids), _i64(d_ids), _f32(idf)
    B, Q = q_ids.shape
    L = d_ids.shape[1]
    edges = _f32(edges)
    nbins = edges.shape[0]
    w1, b1, w2, b2 = _f32(w1), _f32(b1), _f32(w2).reshape(-1), _f32(b2).reshape(-1)
    nodes = w1.shape[0]
    gate_w, out_w, out_b = _f32(gate_w).reshape(-1), _f32(out_w).reshape(-1), _f32(out_b).reshape(-1)
    emb_raw = None if emb_raw is None else _f32(emb_raw)
    out = np.empty(B, dtype=np.float32)
    counts = np.empty((B, Q, nbins + 1), dtype=np.int32)
    err = lib().oracle_drmm(_p(q_ids), _p(d_ids), _p(idf), B, Q, L, _p(packed), ctypes.c_int64(packed.shape[0]), D,
                            _p(edges), nbins, HIST_TYPES[hist_type], GATE_TYPES[gate_type], _p(gate_w), _p(emb_raw),
                            ctypes.c_int64(D), _p(w1), _p(b1), nodes, _p(w2), _p(b2), _p(out_w), _p(out_b), _p(out),
                            _p(counts))
    return out, counts, err


def drmm_from_counts(counts, q_ids, idf, V, D, hist_type, gate_type, gate_w, emb_raw, w1, b1, w2, b2, out_w, out_b):
    """The back end of `drmm` (histogram type -> ffw -> gate -> output layer) on given raw bin counts [B, Q, nbins + 1]."""
    q_ids, idf = _i64(q_ids), _f32(idf)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    B, Q, NB = counts.shape
    w1, b1, w2, b2 = _f32(w1), _f32(b1), _f32(w2).reshape(-1), _f32(b2).reshape(-1)
    gate_w, out_w, out_b = _f32(gate_w).reshape(-1), _f32(out_w).reshape(-1), _f32(out_b).reshape(-1)
    emb_raw = None if emb_raw is None else _f32(emb_raw)
    out = np.empty(B, dtype=np.float32)
    err = lib().oracle_drmm_from_counts(_p(counts), _p(q_ids), _p(idf), B, Q, ctypes.c_int64(V), D, NB - 1, HIST_TYPES[hist_type],
                                        GATE_TYPES[gate_type], _p(gate_w), _p(emb_raw), ctypes.c_int64(D), _p(w1), _p(b1), w1.shape[0],
                                        _p(w2), _p(b2), _p(out_w), _p(out_b), _p(out))
    return out, err


def drmmtks(q_ids, d_ids, idf, packed, D, topk, gate_w, ffw_w, ffw_b, out_w, out_b):
    q_ids, d_ids, idf = _i64(q_ids), _i64(d_ids), _f32(idf)
    B, Q = q_ids.shape
    L = d_ids.shape[1]
    gate_w, ffw_w, ffw_b, out_w, out_b = (_f32(x).reshape(-1) for x in (gate_w, ffw_w, ffw_b, out_w, out_b))
    out = np.empty(B, dtype=np.float32)
    err = lib().oracle_drmmtks(_p(q_ids), _p(d_ids), _p(idf), B, Q, L, _p(packed), ctypes.c_int64(packed.shape[0]), D, int(topk),
                               _p(gate_w), _p(ffw_w), _p(ffw_b), _p(out_w), _p(out_b), _p(out))
    return out, err


NONLIN = {"none": 0, "relu": 1, "tanh": 2}


def pacrr(q_ids, d_ids, idf, packed, D, mingram, maxgram, nfilters, kmax, conv_ws, conv_bs, use_idf, w1, b1, w2, b2, w3, b3, nonlinearity="relu"):
    """conv_ws: list of [nfilters, 1, ng, ng] arrays (ng = mingram..maxgram); conv_bs: list of [nfilters]."""
    q_ids, d_ids, idf = _i64(q_ids), _i64(d_ids), _f32(idf)
    B, Q = q_ids.shape
    L = d_ids.shape[1]
    cw = np.ascontiguousarray(np.concatenate([_f32(w).reshape(-1) for w in conv_ws]))
    cb = np.ascontiguousarray(np.concatenate([_f32(b).reshape(-1) for b in conv_bs]))
    w1, b1, w2, b2, w3, b3 = (_f32(x) for x in (w1, b1, w2, b2, w3, b3))
    C = w1.shape[0]
    out = np.empty(B, dtype=np.float32)
    err = lib().oracle_pacrr(_p(q_ids), _p(d_ids), _p(idf), B, Q, L, _p(packed), ctypes.c_int64(packed.shape[0]), D, int(mingram), int(maxgram),
                             int(nfilters), int(kmax), _p(cw), _p(cb), int(bool(use_idf)), int(C), _p(w1), _p(b1), _p(w2), _p(b2),
                             _p(w3.reshape(-1)), _p(b3.reshape(-1)), NONLIN[nonlinearity], _p(out))
    return out, err


def convknrm(q_ids, d_ids, emb, conv_ws, conv_bs, crossmatch, mu, sigma, w1, b1, w2=None, b2=None, score_tanh=False):
    """conv_ws: list of Conv1d weights [F, D, g] for g = 1..G; conv_bs: list of [F].  w2 is None: single combine layer."""
    q_ids, d_ids, emb = _i64(q_ids), _i64(d_ids), _f32(emb)
    B, Q = q_ids.shape
    L = d_ids.shape[1]
    V, D = emb.shape
    G, F = len(conv_ws), conv_ws[0].shape[0]
    cw = np.ascontiguousarray(np.concatenate([_f32(w).reshape(-1) for w in conv_ws]))
    cb = np.ascontiguousarray(np.concatenate([_f32(b).reshape(-1) for b in conv_bs]))
    mu, sigma, w1, b1 = _f32(mu), _f32(sigma), _f32(w1), _f32(b1)
    H = 0 if w2 is None else w1.shape[0]
    w2 = None if w2 is None else _f32(w2).reshape(-1)
    b2 = None if b2 is None else _f32(b2).reshape(-1)
    out = np.empty(B, dtype=np.float32)
    err = lib().oracle_convknrm(_p(q_ids), _p(d_ids), B, Q, L, _p(emb), ctypes.c_int64(V), D, G, F, _p(cw), _p(cb), int(bool(crossmatch)),
                                _p(mu), _p(sigma), mu.shape[0], _p(w1.reshape(-1)), _p(b1.reshape(-1)), H, _p(w2), _p(b2), int(bool(score_tanh)),
                                _p(out))
    return out, err
