"""Every order 9..16 of every register-resident large-order kernel (and of the LDS-resident
fallbacks behind them), both dtypes, against the oracle on thousands of matrices: these
kernels live far beyond 256 registers per lane, where a toolchain slip shows up as wrong
values in SOME lanes, so small batches are not enough."""
import numpy as np
import pytest
import torch
from conftest import TOL, relerr

pytestmark = pytest.mark.gpu


def N():
    import nitorch_fastmath_amd as N_
    return N_


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def spd_np(n, M, dtype, seed):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, M, M))
    A = G @ G.transpose(0, 2, 1) / M + np.eye(M)
    iu = [(i, j) for i in range(M) for j in range(i + 1, M)]
    c = np.concatenate([np.stack([A[:, i, i] for i in range(M)], -1), np.stack([A[:, i, j] for i, j in iu], -1)], -1)
    return c.astype(dtype), rng.standard_normal((n, M)).astype(dtype)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', range(9, 17))
def test_sym_large_orders(dev, oracle, dn, M):
    dtype = np.float32 if dn == 'f32' else np.float64
    n = 3000 + M            # several tiles + a ragged tail
    mat, vec = spd_np(n, M, dtype, 900 + M)
    S = N().sym
    for rep in range(2):     # twice: a lane-dependent slip is rarely identical run to run
        assert relerr(S.sym_solve(t(mat, dev), t(vec, dev)).cpu().numpy(), oracle.sym_solve(mat, vec)) <= TOL[dn]
        assert relerr(S.sym_invert(t(mat, dev)).cpu().numpy(), oracle.sym_invert(mat)) <= TOL[dn]
        assert relerr(S.sym_invert(t(mat, dev), diag=True).cpu().numpy(), oracle.sym_invert(mat, diag=True)) <= TOL[dn]
        assert relerr(S.sym_det(t(mat, dev)).cpu().numpy(), oracle.sym_det(mat)) <= TOL[dn] * 4
        assert np.array_equal(S.sym_matvec(t(mat, dev), t(vec, dev)).cpu().numpy(), oracle.sym_matvec(mat, vec))
    # strided operands take the LDS-resident fallback: same answers
    ms = t(mat, dev).t().contiguous().t()
    assert relerr(S.sym_solve(ms, t(vec, dev)).cpu().numpy(), oracle.sym_solve(mat, vec)) <= TOL[dn]
    assert relerr(S.sym_invert(ms).cpu().numpy(), oracle.sym_invert(mat)) <= TOL[dn]


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', range(9, 17))
def test_general_large_orders(dev, oracle, dn, n):
    dtype = np.float32 if dn == 'f32' else np.float64
    nb = 2000 + n
    rng = np.random.default_rng(70 + n)
    a = (rng.standard_normal((nb, n, n)) + 8 * np.eye(n)).astype(dtype)
    a[::7, 0, 0] = 0          # force row exchanges in some lanes only (divergent pivoting)
    B = N().batched
    ref_inv, ref_det = oracle.batch_inv(a), oracle.batch_det(a)
    easy = np.ones(nb, bool)
    easy[::7] = False         # the zero-pivot matrices are worse conditioned: looser bound
    hard_tol = 2e-2 if dn == 'f32' else 1e-8   # garbage detector, not a precision claim

    def check_inv(got):
        assert relerr(got[easy], ref_inv[easy]) <= TOL[dn]
        assert relerr(got[~easy], ref_inv[~easy]) <= hard_tol
        eye = np.einsum('bij,bjk->bik', a.astype(np.float64), got.astype(np.float64))
        assert np.abs(eye - np.eye(n)).max() <= hard_tol

    for rep in range(2):
        check_inv(B.batchinv(t(a, dev)).cpu().numpy())
        d = B.batchdet(t(a, dev)).cpu().numpy()
        assert relerr(d[easy], ref_det[easy]) <= TOL[dn] * 4 and relerr(d[~easy], ref_det[~easy]) <= hard_tol
    at = t(a.transpose(0, 2, 1).copy(), dev).transpose(-1, -2)
    check_inv(B.batchinv(at).cpu().numpy())
