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


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rand_rboxes(rng, n, span=1024.0, lo=4.0, hi=100.0):
    """SURVEY.md 8(d) config-1 box distribution."""
    b = np.empty((n, 5), np.float32)
    b[:, 0:2] = rng.uniform(0, span, (n, 2))
    b[:, 2:4] = rng.uniform(lo, hi, (n, 2))
    b[:, 4] = rng.uniform(-np.pi / 4, 3 * np.pi / 4, n)
    return b


def distinct_scores(rng, n):
    return ((rng.permutation(n).astype(np.float32) + 1) / np.float32(n + 1) * np.float32(0.95)
            + np.float32(0.05))


@pytest.fixture
def rng():
    return np.random.default_rng(1234)
