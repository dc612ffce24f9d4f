/* Host-side glue of `PytorchTrainer.predict` (capreolus_amd/trainer/pytorch.py; reference trainer/pytorch.py:342-348): the
 * {qid: {docid: score}} dictionaries of a scored run, built from the fp16 score vector the kernels hand back.
 *
 * Not part of the C-ABI of include/capreolus_amd.h (that boundary has no Python types): a CPython helper, loaded with ctypes.PyDLL so
 * that the GIL stays held, compiled by csrc/build.py with gcc against the interpreter's own headers.  It does what
 *     {qid: dict(zip(docids, scores_f16[lo:lo + len(docids)].tolist())) for qid, docids, lo in groups}
 * does - the same keys, the same insertion order, the same values (float(np.float16(x)): the reference rounds through float16
 * before the scores reach the run file) - without creating one zip tuple, one list slot and one float object per pair: the 65,536
 * possible fp16 values exist ONCE as float objects (`lut`), a pair costs one table lookup and one insert into a presized dict.
 * At 64,000 pairs that is what `predict` spends its time on once the kernels take 0.6 ms.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

/* groups: list of (qid, tuple of docids, offset of the group's first pair in the run); groups[g0:g1] are converted.
 * bits: fp16 bit patterns of the pairs `base` .. `base + n - 1` of the run.  lut: tuple of 65,536 floats.  out: dict to fill.
 * `merge` != 0: a qid may come in several runs of samples - later groups update the dict of earlier ones (dict.update semantics).
 * Returns 0, or -1 with a Python exception set. */
int capamd_preds_from_fp16(PyObject* groups, Py_ssize_t g0, Py_ssize_t g1, const uint16_t* bits, Py_ssize_t base, Py_ssize_t n, PyObject* lut,
                           PyObject* out, int merge) {
  if (!PyList_Check(groups) || !PyTuple_Check(lut) || PyTuple_GET_SIZE(lut) != 65536 || !PyDict_Check(out) || !bits) {
    PyErr_SetString(PyExc_TypeError, "capamd_preds_from_fp16: groups must be a list, lut a tuple of 65536 floats, out a dict");
    return -1;
  }
  if (g0 < 0 || g1 > PyList_GET_SIZE(groups) || g0 > g1) {
    PyErr_SetString(PyExc_IndexError, "capamd_preds_from_fp16: group range");
    return -1;
  }
  PyObject* const* lv = &PyTuple_GET_ITEM(lut, 0);
  for (Py_ssize_t g = g0; g < g1; ++g) {
    PyObject* grp = PyList_GET_ITEM(groups, g);
    if (!PyTuple_Check(grp) || PyTuple_GET_SIZE(grp) < 3 || !PyTuple_Check(PyTuple_GET_ITEM(grp, 1))) {
      PyErr_SetString(PyExc_TypeError, "capamd_preds_from_fp16: a group is (qid, tuple of docids, offset)");
      return -1;
    }
    PyObject* qid = PyTuple_GET_ITEM(grp, 0);
    PyObject* docids = PyTuple_GET_ITEM(grp, 1);
    const Py_ssize_t lo = PyLong_AsSsize_t(PyTuple_GET_ITEM(grp, 2));
    if (lo == -1 && PyErr_Occurred()) return -1;
    const Py_ssize_t cnt = PyTuple_GET_SIZE(docids);
    if (lo < base || lo + cnt > base + n) {
      PyErr_SetString(PyExc_IndexError, "capamd_preds_from_fp16: a group reaches outside the score vector");
      return -1;
    }
    PyObject* d = NULL;
    int fresh = 1;
    if (merge) {
      d = PyDict_GetItemWithError(out, qid);   /* borrowed */
      if (!d && PyErr_Occurred()) return -1;
      if (d) { Py_INCREF(d); fresh = 0; }
    }
    if (!d) d = _PyDict_NewPresized(cnt);
    if (!d) return -1;
    const uint16_t* b = bits + (lo - base);
    for (Py_ssize_t i = 0; i < cnt; ++i) {
      if (PyDict_SetItem(d, PyTuple_GET_ITEM(docids, i), lv[b[i]]) < 0) { Py_DECREF(d); return -1; }
    }
    if (fresh && PyDict_SetItem(out, qid, d) < 0) { Py_DECREF(d); return -1; }
    Py_DECREF(d);
  }
  return 0;
}
