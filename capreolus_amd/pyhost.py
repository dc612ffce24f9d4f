"""The {qid: {docid: score}} dictionaries of a scored run, built by csrc/pyhost.c (a CPython helper loaded with ctypes.PyDLL).

`preds_from_fp16(groups, scores_f16, out)` equals, key for key and in the same insertion order,

    for qid, docids, lo in groups: out[qid] = dict(zip(docids, scores_f16[lo:lo + len(docids)].tolist()))

(reference trainer/pytorch.py:346-348: the scores reach the predictions rounded through float16).  The 65,536 possible values are
`float` objects made once; a pair costs a table lookup and an insert into a presized dict instead of a zip tuple, a list slot and a
new float - dict building is what `predict` spends its time on once the kernels take 0.6 ms per 64,000 pairs.  This is host glue,
not scoring: when the helper was not built (csrc/build.py builds it with gcc) the same dictionaries come from the Python expression.
"""
import ctypes
import os

import numpy as np

import sysconfig

# (named with the interpreter's ABI tag by csrc/build.py: a helper built for another Python - it uses CPython's object layout - is not found)
_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libcapamd_pyhost" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
_fn = None
_lut = None


def _load():
    global _fn, _lut
    if _fn is None:
        if not os.path.exists(_PATH):
            _fn = False
            return _fn
        try:
            fn = ctypes.PyDLL(_PATH).capamd_preds_from_fp16
        except (OSError, AttributeError):      # an unloadable or foreign file: the Python expression below gives the same dictionaries
            _fn = False
            return _fn
        fn.argtypes = [ctypes.py_object, ctypes.c_ssize_t, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_ssize_t, ctypes.py_object,
                       ctypes.py_object, ctypes.c_int]
        fn.restype = ctypes.c_int       # -1: a Python exception is set (PyDLL re-raises it)
        _lut = tuple(np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float64).tolist())
        _fn = fn
    return _fn


def available():
    return bool(_load())


def preds_from_fp16(groups, scores_f16, out, g0=0, g1=None, base=0, merge=False):
    """Adds groups[g0:g1] (a list of (qid, tuple of docids, offset of the group's first pair in the run)) to `out`; `scores_f16` is a
    contiguous numpy float16 vector holding the run's pairs base .. base + len - 1.  merge: a qid may come in several groups (later
    groups update the earlier dict, as dict.update would)."""
    g1 = len(groups) if g1 is None else g1
    if scores_f16.dtype != np.float16 or not scores_f16.flags.c_contiguous:
        raise TypeError("scores_f16 must be a contiguous float16 vector")
    fn = _load()
    if fn:
        fn(groups, g0, g1, scores_f16.ctypes.data, base, scores_f16.size, _lut, out, int(bool(merge)))
        return out
    for qid, docids, lo in groups[g0:g1]:
        vals = scores_f16[lo - base:lo - base + len(docids)].tolist()
        if merge and qid in out:
            out[qid].update(zip(docids, vals))
        else:
            out[qid] = dict(zip(docids, vals))
    return out
