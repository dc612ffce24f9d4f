"""ctypes binding of the C-ABI library (include/capreolus_amd.h).

There is no fallback: if csrc/libcapreolus_amd.so is missing or does not export every declared
symbol the import of the scoring engine fails loudly (the reference's prediction path also fails
loudly, capreolus/sampler/__init__.py:230-233).
"""
import ctypes
import os

# Load PyTorch's HIP runtime FIRST.  The ROCm wheels bundle their own libamdhip64.so.7 (same SONAME
# as /opt/rocm's); if this library were dlopen'ed before torch, the process would end up with two HIP
# runtimes and kernels registered with one could not be launched on streams created by the other.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CAPAMD_LIB_PATH") or os.path.join(_HERE, "csrc", "libcapreolus_amd.so")  # override: profiling builds only

OK, ERR_ARG, ERR_ALIGN, ERR_LAUNCH, ERR_WORKSPACE = 0, 1, 2, 3, 4
LAUNCH_CONCURRENT = 1
STATUS_DOC_ID_RANGE, STATUS_QUERY_ID_RANGE, STATUS_QUERY_OOV, STATUS_SCORE_NAN, STATUS_TIE_RANGE, STATUS_LIST_QUERY = 1, 2, 4, 8, 16, 32

_vp, _i, _i64, _sz, _u, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_uint, ctypes.c_float



class BertModel(ctypes.Structure):
    """capamd_bert_model (include/capreolus_amd.h)."""

    _fields_ = [(n, ctypes.c_int) for n in ("hidden", "layers", "heads", "ffn", "vocab", "max_pos", "type_vocab", "compute_dtype")] + [
        (n, ctypes.c_void_p) for n in ("word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b", "pooler_w", "pooler_b",
                                       "cls_w", "cls_b", "blob", "layer_f32")] + [("ln_eps", ctypes.c_float), ("pos_pad_id", ctypes.c_int)]


_mp = ctypes.POINTER(BertModel)

# name -> (restype, argtypes); must list every symbol include/capreolus_amd.h declares
SIGNATURES = {
    "capamd_version": (_i, []),
    "capamd_arch": (ctypes.c_char_p, []),
    "capamd_interaction_workspace_bytes": (_sz, []),
    "capamd_packed_row_stride": (_i64, [_i]),
    "capamd_packed_table_bytes": (_i64, [_i64, _i]),
    "capamd_pack_embeddings": (_i, [_vp, _i64, _i, _i64, _vp, _vp]),
    "capamd_similarity_matrix": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _i, _vp, _vp, _vp]),
    "capamd_knrm_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _u, _vp]),
    "capamd_lists_workspace_bytes": (_sz, [_i, _i64, _i64, _i]),
    "capamd_lists_workspace_bytes_q": (_sz, [_i, _i64, _i64, _i, _i]),
    "capamd_knrm_forward_lists": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "capamd_drmm_forward_lists": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _vp, _i, _i, _i, _vp, _vp, _i64, _vp, _vp, _i, _vp, _vp,
                                       _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "capamd_drmmtks_forward_lists": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                          _vp]),
    "capamd_drmmtks_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "capamd_drmmtks_features": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "capamd_pacrr_forward_lists": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "capamd_pacrr_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _vp]),
    "capamd_convknrm_table_bytes": (_i64, [_i64, _i, _i]),
    "capamd_convknrm_pack_tables": (_i, [_vp, _i64, _i, _i64, _vp, _vp, _i, _i, _vp, _vp]),
    "capamd_convknrm_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "capamd_convknrm_lists_workspace_bytes": (_sz, [_i, _i64, _i, _i, _i]),
    "capamd_convknrm_forward_lists": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "capamd_knrm_features": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "capamd_drmm_train_step_workspace_floats": (_sz, [_i, _i, _i, _i]),
    "capamd_drmm_train_step": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _vp, _i, _i, _i, _vp, _i, _f, _f, _f, _f, _f, _vp, _vp, _sz, _vp, _vp]),
    "capamd_drmmtks_train_step_workspace_floats": (_sz, [_i, _i, _i]),
    "capamd_drmmtks_train_step": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _i, _vp, _i, _f, _f, _f, _f, _f, _vp, _vp, _sz, _vp, _vp]),
    "capamd_knrm_train_step_workspace_floats": (_sz, [_i, _i]),
    "capamd_knrm_train_step": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _i, _vp, _i, _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _sz, _vp, _vp]),
    "capamd_drmm_features": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "capamd_knrm_forward_indexed": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _u,
                                         _vp]),
    "capamd_drmm_forward_indexed": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _vp, _i, _i, _i, _vp, _vp, _i64, _vp, _vp, _i,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _u, _vp]),
    "capamd_drmm_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _vp, _i, _i, _i, _vp, _vp, _i64, _vp, _vp, _i,
                                 _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _u, _vp]),
    "capamd_bert_blob_bytes": (_i64, [_mp]),
    "capamd_bert_layer_f32_floats": (_i64, [_mp]),
    "capamd_bert_pack_layer": (_i, [_mp, _i, ctypes.POINTER(_vp), _vp, _vp, _vp]),
    "capamd_bert_workspace_bytes": (_i64, [_mp, _i, _i64, _i64]),
    "capamd_bert_maxp_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _mp, _i, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "capamd_cedr_passage_features": (_i, [_vp, _vp, _vp, _i, _i, _i, _mp, _i64, _vp, _i64, _i, _vp, ctypes.POINTER(_i), _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "capamd_cedr_score": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "capamd_maxp_pool": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "capamd_bert_gemm": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "capamd_bert_gemm_ln": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "capamd_ngram_conv_workspace_floats": (_sz, [_i, _i, _i, _i]),
    "capamd_ngram_conv_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "capamd_ngram_conv_backward": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _i, _i, _i, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _sz, _vp, _vp]),
    "capamd_convknrm_train_step_workspace_floats": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "capamd_convknrm_train_step": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _i, _i, _i, _i, _i, ctypes.POINTER(_vp), _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _sz, _vp, _vp]),
    "capamd_pacrr_train_step_workspace_floats": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "capamd_pacrr_train_step": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, ctypes.POINTER(_vp), _i, _f, _f, _f, _f, _f, _vp, _vp, _sz,
                                     _vp, _vp]),
    "capamd_kernel_pool_chunks": (_i, [_i]),
    "capamd_kernel_pool_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "capamd_kernel_pool_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "capamd_pacrr_convmax_forward": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "capamd_pacrr_convmax_backward": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "capamd_rank_candidates": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "capamd_ndcg_cut": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "capamd_bert_qkv_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
}

# builder-side profiling hooks (capreolus_amd/csrc/capamd_profiling.h): exported ONLY by the -DCAPAMD_PROFILING build of the library
# (csrc/libcapreolus_amd_prof.so), never by the product library - see profiling_build()
PROF_LIB_PATH = os.environ.get("CAPAMD_PROF_LIB_PATH") or os.path.join(_HERE, "csrc", "libcapreolus_amd_prof.so")  # override: ablation builds only
PROFILING_SIGNATURES = {
    "capamd_debug_set_gemm_stamps": (None, [_vp]),
    "capamd_debug_ffn1_timing": (None, [_i]),
    "capamd_debug_ffn1_timing_read": (_i, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "capamd_debug_lists_timing": (None, [_i]),
    "capamd_debug_lists_timing_read": (_i, [ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


class EngineError(RuntimeError):
    """A C-ABI call returned a non-zero CAPAMD_ERR_* code."""


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP scoring library has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or capreolus_amd/csrc/build.py). "
            "There is no CPU fallback for the scoring path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


_prof = None


class profiling_build:
    """`with _lib.profiling_build() as lib:` - inside the block every engine call goes to the -DCAPAMD_PROFILING build of the library
    (the same kernels plus the event hooks of csrc/capamd_profiling.h); `lib` has the hooks bound.  For bench.py's per-pass timing
    legs and the scripts/ probes: the product library has no such hooks and no mutable global state."""

    def __enter__(self):
        global _lib, _prof
        load()
        if _prof is None:
            if not os.path.exists(PROF_LIB_PATH):
                raise ImportError(f"{PROF_LIB_PATH} is missing (capreolus_amd/csrc/build.py builds it next to the product library)")
            lib = ctypes.CDLL(PROF_LIB_PATH)
            for name, (res, args) in list(SIGNATURES.items()) + list(PROFILING_SIGNATURES.items()):
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            _prof = lib
        self.saved, _lib = _lib, _prof
        return _prof

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


_ERR_NAMES = {ERR_ARG: "CAPAMD_ERR_ARG", ERR_ALIGN: "CAPAMD_ERR_ALIGN", ERR_LAUNCH: "CAPAMD_ERR_LAUNCH",
              ERR_WORKSPACE: "CAPAMD_ERR_WORKSPACE"}


def check(rc, what):
    if rc != OK:
        raise EngineError(f"{what} failed with {_ERR_NAMES.get(rc, rc)}")
