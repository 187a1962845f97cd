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
def ctx():
    from mx_deepim_amd.runtime import Context
    return Context.get(0)


@pytest.fixture(scope="session")
def small_batch():
    """2 synthetic 480x640 pairs (config 1 shape, B=2), built once."""
    from mx_deepim_amd import synthetic
    return synthetic.make_batch(2, seed=2333, n_frames=2)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-30, np.max(np.abs(b))))
