"""Stub-import harness for the *reference* rerankers (generator side only).

This file is used ONLY by tests/golden/make_golden.py, in the build container,
where /root/reference exists.  It never runs on the GPU box and nothing under
tests/ imports it at test time.  It does not copy reference code: it only
arranges for `capreolus.reranker.{common,KNRM,DRMM,ptBERTMaxP}` to be importable
from /root/reference without the packages the container lacks (profane,
tensorflow, tensorflow_ranking, jnius, matplotlib is present) by pre-seeding
sys.modules with empty stand-ins for *third-party* modules.  The reference
tree is never written to (PYTHONDONTWRITEBYTECODE is forced).
"""
import importlib
import importlib.machinery
import logging
import os
import sys
import types

REF_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    """Returns (common, KNRM, DRMM, ptBERTMaxP, DRMMTKS, PACRR, ConvKNRM, CEDRKNRM) reference modules."""
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present; golden fixtures can only be regenerated in the build container")

    import torch  # noqa: F401  (real)
    import transformers  # noqa: F401  (real; must be imported before tensorflow is stubbed)
    from transformers import AutoModelForSequenceClassification  # noqa: F401

    class ConfigOption:
        def __init__(self, key, default_value=None, description=None, value_type=None):
            self.key, self.default_value = key, default_value

    class Dependency:
        def __init__(self, key, module, name=None, default_config_overrides=None):
            self.key, self.module, self.name = key, module, name

    class ModuleBase:
        @classmethod
        def register(cls, sub):
            return sub

    cap = _stub("capreolus", ConfigOption=ConfigOption, Dependency=Dependency, ModuleBase=ModuleBase,
                get_logger=lambda name=None: logging.getLogger(name or "capreolus"))
    cap.__path__ = [os.path.join(REF_ROOT, "capreolus")]

    class Reranker(ModuleBase):
        pass

    rr = _stub("capreolus.reranker", Reranker=Reranker)
    rr.__path__ = [os.path.join(REF_ROOT, "capreolus", "reranker")]
    cap.reranker = rr
    ut = _stub("capreolus.utils")
    ut.__path__ = [os.path.join(REF_ROOT, "capreolus", "utils")]
    _stub("capreolus.utils.loginit", get_logger=lambda name=None: logging.getLogger(name or "capreolus"))

    class _Layer:  # tensorflow.keras.layers.Layer stand-in (TF twins are never instantiated)
        def __init__(self, *a, **k):
            pass

    class _Model:
        def __init__(self, *a, **k):
            pass

    keras = _stub("tensorflow.keras", Model=_Model)
    _stub("tensorflow", keras=keras, Variable=None, float32=None)
    layers = _stub("tensorflow.keras.layers", Layer=_Layer)
    keras.layers = layers
    _stub("tensorflow.python")
    _stub("tensorflow.python.keras")
    _stub("tensorflow.python.keras.losses", CategoricalCrossentropy=object)
    _stub("tensorflow_ranking")
    _stub("tensorflow_ranking.python")
    _stub("tensorflow_ranking.python.keras")
    _stub("tensorflow_ranking.python.keras.losses", PairwiseHingeLoss=object)

    mods = [importlib.import_module("capreolus.reranker." + n) for n in ("common", "KNRM", "DRMM", "ptBERTMaxP", "DRMMTKS", "PACRR", "ConvKNRM", "CEDRKNRM")]
    return tuple(mods)
