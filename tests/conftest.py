import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


@pytest.fixture(params=golden_names())
def golden(request):
    return load_golden(request.param)


def rel_fro(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (den if den > 0 else 1.0))


@pytest.fixture
def lib_options():
    """set developer switches of libganq_hip.so for one test (ganq_debug_set_option); every touched option is reset to its
    default afterwards.  lib_options(reset=(names...), NAME=value, ...)"""
    from ganq_amd import _lib

    touched = set()

    def apply(reset=(), **opts):
        for name in reset:
            _lib.debug_option(name, None)
            touched.add(name)
        for name, value in opts.items():
            _lib.debug_option(name, value)
            touched.add(name)

    yield apply
    for name in touched:
        _lib.debug_option(name, None)
