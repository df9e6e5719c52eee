import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def f3d():
    """The product package (directory name carries a hyphen, so it is imported through importlib)."""
    return importlib.import_module("cuda-flow3d_amd")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); builds oracle/liboracle.so on first use."""
    from oracle import oracle as orc
    orc.lib()
    return orc


def box_in_container(rng, dims, cdims, lo=-1.0, hi=1.0, poison=True):
    """Random [d,h,w] box in the corner of a NaN-poisoned [Dc,Hc,Wc] host container."""
    w, h, d = dims
    wc, hc, dc = cdims
    c = np.full((dc, hc, wc), np.nan if poison else 0.0, np.float32)
    c[:d, :h, :w] = rng.uniform(lo, hi, size=(d, h, w)).astype(np.float32)
    return c


def same(a, b):
    """Exact equality that treats -0 == +0 and forbids NaN in either array."""
    a = np.asarray(a)
    b = np.asarray(b)
    return a.shape == b.shape and not np.isnan(a).any() and not np.isnan(b).any() and bool(np.all(a == b))


def bit_same(a, b):
    return a.shape == b.shape and bool(np.all(np.ascontiguousarray(a).view(np.uint32) == np.ascontiguousarray(b).view(np.uint32)))


def flush_c_stdio():
    """The native side prints with printf; push libc's buffers out so capfd sees the text."""
    import ctypes
    ctypes.CDLL(None).fflush(None)
