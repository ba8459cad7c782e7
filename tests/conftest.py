import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (built on demand with gcc)."""
    import oracle as _oracle
    _oracle.build()
    return _oracle


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test is marked gpu but no ROCm device is visible")
    return torch.device("cuda:0")


def unit_ball_clouds(rng, B, N):
    """Synthetic clouds as SURVEY.md 8(d): random direction x U^(1/3) radius, centred and
    divided by the max norm (the normalisation of reference src/datasets/modelnet.py:66-69)."""
    import numpy as np
    v = rng.standard_normal((B, N, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    r = rng.random((B, N, 1)) ** (1.0 / 3.0)
    p = v * r
    if N < 4:  # a single point has no extent to normalise by
        return p.astype(np.float32)
    p = p - p.mean(axis=1, keepdims=True)
    p = p / np.sqrt((p ** 2).sum(-1)).max(axis=1)[:, None, None]
    return p.astype(np.float32)
