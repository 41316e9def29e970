import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_fe():
    return np.load(os.path.join(GOLDEN, "frontend_golden.npz"))


@pytest.fixture(scope="session")
def golden_model():
    return np.load(os.path.join(GOLDEN, "b3mtl_golden.npz"))


@pytest.fixture(scope="session")
def clips4():
    from sm_hpss_mtl_amd.synth import synth_clips
    return synth_clips(4, seed=0)


def checks(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a.min(), a.max()])
