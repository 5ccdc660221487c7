"""Skeletal-namespace loader for the upstream reference (build container only).

Loads ``gptqmodel/quantization/{config,quantizer,gptq,ganq}.py`` and
``gptqmodel/looper/named_module.py`` straight from ``/root/reference`` WITHOUT executing
any package ``__init__`` (the package itself does not import under transformers 5.x, see
SURVEY.md section 8c).  Used by ``make_golden.py`` to produce the committed fixtures and by
nothing else: ``/root/reference`` does not exist on the GPU box, and nothing in the product,
the ``-m gpu`` tests, ``smoke()`` or ``bench.py`` imports this file.

The reference depends on a few packages that are not installed here; they are replaced by
inert stand-ins (loggers, Apple-MLX probes).  ``kmeans1d`` (un-vendored third-party C++
dependency, pinned at smpanaro/kmeans1d@831c169c in the reference's requirements.txt:16) is
replaced by a callable the caller supplies, so the golden vectors record the initial
codebook T0 as an *input* (parity of T0 itself is unpinned).
"""
import os
import sys
import types

REF_ROOT = os.environ.get("GANQ_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "gptqmodel", "quantization"))


class _NullLog:
    """Stands in for logbar.LogBar.shared(): every attribute is a no-op callable that also
    has no-op attributes (the reference calls e.g. ``log.info.once``)."""

    def __getattr__(self, name):
        return _NullLog()

    def __call__(self, *a, **k):
        return None


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def load_reference(kmeans_cluster=None):
    """Return (ganq_module, gptq_module, config_module, NamedModule).

    kmeans_cluster(values[n,1] or [n], k, weights=) -> (labels, centroids) replaces
    kmeans1d.cluster (ganq.py:29).
    """
    if not reference_available():
        raise RuntimeError(f"reference not present at {REF_ROOT}")
    import transformers  # noqa: F401  (must be imported before the mlx stand-in exists)
    import transformers.pytorch_utils  # noqa: F401
    import torch  # noqa: F401

    if "gptqmodel.quantization.ganq" in sys.modules:
        km = sys.modules["kmeans1d"]
        if kmeans_cluster is not None:
            km.cluster = kmeans_cluster
        return (sys.modules["gptqmodel.quantization.ganq"], sys.modules["gptqmodel.quantization.gptq"],
                sys.modules["gptqmodel.quantization.config"], sys.modules["gptqmodel.looper.named_module"].NamedModule)

    class _LogBar:
        @staticmethod
        def shared():
            return _NullLog()

    _stub("logbar", LogBar=_LogBar)
    _stub("tokenicer")
    _stub("device_smi")

    def _no_kmeans(*a, **k):
        raise RuntimeError("kmeans1d is not installed; pass kmeans_cluster= to load_reference")

    _stub("kmeans1d", cluster=kmeans_cluster or _no_kmeans)

    # ganq.py:11-15 wraps the mlx import in try/except ImportError; leaving mlx absent makes
    # USE_MLX False, but ganq.py:32,36,330,358 use mx.array / @mx.compile at import time, so a
    # stand-in with those two attributes is required.
    mlx = _stub("mlx")
    mlx.__path__ = []
    core = _stub("mlx.core", array=type("array", (), {}), compile=lambda f=None, **k: f)
    mlx.core = core

    pkg_root = os.path.join(REF_ROOT, "gptqmodel")
    for name, sub in [("gptqmodel", ""), ("gptqmodel.utils", "utils"), ("gptqmodel.looper", "looper"),
                      ("gptqmodel.adapter", "adapter"), ("gptqmodel.quantization", "quantization")]:
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(pkg_root, sub) if sub else pkg_root]
        pkg.__package__ = name
        sys.modules[name] = pkg

    import importlib

    cfg = importlib.import_module("gptqmodel.quantization.config")
    sys.modules["gptqmodel.quantization"].QuantizeConfig = cfg.QuantizeConfig
    named = importlib.import_module("gptqmodel.looper.named_module")
    gptq = importlib.import_module("gptqmodel.quantization.gptq")
    ganq = importlib.import_module("gptqmodel.quantization.ganq")
    assert ganq.USE_MLX is False
    return ganq, gptq, cfg, named.NamedModule
