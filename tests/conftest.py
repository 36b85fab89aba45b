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


# ---------------------------------------------------------------------------------------------
# Error model for the float32 results that are NOT closed forms (QR family beyond the orders the
# reference itself runs, LU-based solves / inverses of orders > 4): two correct float32
# computations of the same quantity differ by the SUM of their errors, so "within 1e-6 of the
# reference" cannot be asked of them (the reference's own Householder reduction is 9e-6 away from
# the exact result at n = 12).  What is asked instead, against an fp64 TRUTH computed on the same
# inputs (the float64 oracle or numpy.linalg in float64):
#       err(got) <= 2 * err(reference) + c * n * eps        (batch max-norm, relerr above)
# i.e. never materially worse than the reference's own arithmetic, with c * n * eps the rounding
# floor of an O(n)-deep float32 computation.  Closed forms and the orders n <= 5 where the
# reference path itself runs are held to TOL directly.  float64 results are always held to TOL
# (their n * eps floor is 1e-15).
EPS = {'f32': 2.0 ** -23, 'f64': 2.0 ** -52}


def model_bound(ref, truth, n, dn, c=4.0):
    return 2.0 * relerr(ref, truth) + c * n * EPS[dn]


def within_model(got, ref, truth, n, dn, c=4.0):
    """`got` (the kernel, or the oracle) is no worse against `truth` than the error model allows,
    `ref` being the reference-side result (a golden vector, or the oracle)."""
    return relerr(got, truth) <= model_bound(ref, truth, n, dn, c)


def parity_ok(got, ref, n, dn, truth=None, c=4.0):
    """the bar of the module comment: TOL against the reference for float64 and for orders <= 5;
    the error model (needs `truth`) for float32 beyond"""
    if dn == 'f64' or n <= 5 or truth is None:
        return relerr(got, ref) <= TOL[dn]
    return within_model(got, ref, truth, n, dn, c)
