"""Import harness for the *reference* hot-path modules (container-only test tooling).

TEST INFRASTRUCTURE ONLY. This module imports talmolab/sleap-nn modules straight from
``/root/reference`` (never copied) so that ``oracle/gen_golden.py`` can generate golden
input/output vectors (replayed by ``tests/test_oracle_golden.py``, which is how
``oracle/cpu_ref.py`` is pinned against the real reference).  Nothing under ``sleap_nn_amd/`` (the product) may import this file, and it is
never executed on the GPU box (``/root/reference`` does not exist there).

The reference's package ``__init__`` files pull in loguru / sleap_io / lightning /
omegaconf / torchvision, none of which are installed here.  We register inert stand-in
modules for those *third-party* names and namespace shims for the reference packages so
that only the pure hot-path modules execute (SURVEY.md §8c / Appendix C documents the
recipe).  The reference's own arithmetic runs unmodified.
"""

from __future__ import annotations

import os
import pickle
import sys
import types

REFERENCE_ROOT = os.environ.get("SLEAP_NN_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "sleap_nn"))


class _AnyModule(types.ModuleType):
    """Module whose unknown attributes resolve to fresh dummy classes."""

    def __getattr__(self, key):
        if key.startswith("__"):
            raise AttributeError(key)
        return type(key, (), {})


class _NoLog:
    def __getattr__(self, key):
        return lambda *a, **k: None


_installed = False


def install() -> None:
    """Register stubs + namespace shims. Idempotent."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = _AnyModule(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    stub("loguru", logger=_NoLog())
    stub("omegaconf")
    stub("omegaconf.dictconfig")
    stub("sleap_io")
    stub("sleap_io.io")
    stub("sleap_io.io.skeleton")
    stub("skia")
    stub("cv2")
    tv = [
        "torchvision",
        "torchvision.ops",
        "torchvision.ops.misc",
        "torchvision.utils",
        "torchvision.models",
        "torchvision.models.convnext",
        "torchvision.models.swin_transformer",
        "torchvision.transforms",
        "torchvision.transforms.v2",
        "torchvision.transforms.v2.functional",
    ]
    for n in tv:
        stub(n)
    for n in list(sys.modules):
        if n.startswith("torchvision.") or n.startswith("sleap_io.") or n.startswith("omegaconf."):
            par, _, ch = n.rpartition(".")
            if par in sys.modules:
                setattr(sys.modules[par], ch, sys.modules[n])

    def ns(name, path):
        p = types.ModuleType(name)
        p.__path__ = [path]
        sys.modules[name] = p

    root = os.path.join(REFERENCE_ROOT, "sleap_nn")
    ns("sleap_nn", root)
    ns("sleap_nn.inference", os.path.join(root, "inference"))
    ns("sleap_nn.inference.ops", os.path.join(root, "inference", "ops"))
    ns("sleap_nn.inference.layers", os.path.join(root, "inference", "layers"))
    ns("sleap_nn.inference.layers.backends", os.path.join(root, "inference", "layers", "backends"))
    ns("sleap_nn.data", os.path.join(root, "data"))
    ns("sleap_nn.training", os.path.join(root, "training"))
    _installed = True


class AttrDict(dict):
    """Minimal DictConfig stand-in (attribute + item access, ``**`` unpacking)."""

    __getattr__ = dict.__getitem__


def attrdict(x):
    if isinstance(x, dict):
        return AttrDict({k: attrdict(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return [attrdict(v) for v in x]
    return x


class _Stub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, s):
        self.__dict__["_state"] = s


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, mod, name):
        try:
            return super().find_class(mod, name)
        except Exception:
            return type(name, (_Stub,), {})


def tolerant_pickle_module():
    pm = types.ModuleType("pickle")
    pm.Unpickler = _TolerantUnpickler
    pm.load = lambda f, **k: _TolerantUnpickler(f, **k).load()
    pm.__name__ = "pickle"
    return pm


def load_lightning_ckpt_state(path: str):
    """Return the ``Model`` state dict (``model.`` prefix stripped) of a Lightning ckpt."""
    import torch

    ck = torch.load(path, map_location="cpu", weights_only=False, pickle_module=tolerant_pickle_module())
    sd = ck["state_dict"]
    return {k[len("model.") :]: v for k, v in sd.items() if k.startswith("model.")}


def load_pickle_tolerant(path: str):
    with open(path, "rb") as f:
        return _TolerantUnpickler(f).load()
