import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_sym():
    return np.load(os.path.join(GOLDEN, 'sym.npz'))


@pytest.fixture(scope='session')
def golden_batched():
    return np.load(os.path.join(GOLDEN, 'batched.npz'))


@pytest.fixture(scope='session')
def golden_reduce():
    return np.load(os.path.join(GOLDEN, 'reduce.npz'))


@pytest.fixture(scope='session')
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


# tolerances of BASELINE.json's north_star: 1e-6 rel fp32 / 1e-12 rel fp64, max-norm
TOL = {'f32': 1e-6, 'f64': 1e-12}


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b).max() if b.size else 1.0
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0)) if a.size else 0.0
