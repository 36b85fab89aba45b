"""Every order 9..16 of every register-resident large-order kernel (and of the LDS-resident
fallbacks behind them), both dtypes, against the oracle on thousands of matrices: these
kernels live far beyond 256 registers per lane, where a toolchain slip shows up as wrong
values in SOME lanes, so small batches are not enough."""
import numpy as np
import pytest
import torch
from conftest import TOL, EPS, relerr

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


def per_matrix_err(x, truth):
    """max-norm relative error of every matrix of the batch against its own truth"""
    x, truth = x.astype(np.float64).reshape(len(x), -1), truth.astype(np.float64).reshape(len(truth), -1)
    return np.abs(x - truth).max(-1) / np.maximum(np.abs(truth).max(-1), 1e-300)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', range(9, 17))
def test_general_large_orders(dev, oracle, dn, n):
    """`a.inverse()` / `a.det()` of the reference (`_impl/batched.py:119-120`, `:53-54`) at orders 9..16.

    Six matrices in seven are well conditioned (cond_2 <= ~10): held to TOL against the oracle.
    Every seventh has a ZERO leading pivot (a[0, 0] = 0), which forces a row exchange in some lanes
    only (divergent pivoting) and leaves cond_2 between 1e2 and 1e5 (printed by
    scripts/accuracy_study.py large, profiles/r02/accuracy_large.md): an LU-based inverse is then
    only accurate to ~eps * cond, whoever computes it, so those are held to the error model
        err_i <= 2 * err_oracle_i + n * eps * (1 + cond_i / 8)      per matrix i,
    errors measured against numpy.linalg in float64 on the same input (float64 input: against the
    oracle itself, whose own error is n * eps * cond ~ 1e-11 there)."""
    dtype = np.float32 if dn == 'f32' else np.float64
    nb = 2000 + n
    rng = np.random.default_rng(70 + n)
    a = (rng.standard_normal((nb, n, n)) + 8 * np.eye(n)).astype(dtype)
    a[::7, 0, 0] = 0          # force row exchanges in some lanes only (divergent pivoting)
    B = N().batched
    ref_inv, ref_det = oracle.batch_inv(a), oracle.batch_det(a)
    easy = np.ones(nb, bool)
    easy[::7] = False
    a64 = a.astype(np.float64)
    cond = np.linalg.cond(a64)
    assert cond[easy].max() < 25 and cond[~easy].max() < 1e6
    true_inv, true_det = np.linalg.inv(a64), np.linalg.det(a64)
    floor = n * EPS[dn] * (1 + cond / 8)

    def check(got, ref, truth):
        assert relerr(got[easy], ref[easy]) <= TOL[dn] * (4 if got.ndim == 1 else 1)
        if dn == 'f64':   # no wider truth at hand: the oracle's own error is far below the floor
            assert (per_matrix_err(got, ref) <= floor * 2)[~easy].all()
        else:
            e_got, e_ref = per_matrix_err(got, truth), per_matrix_err(ref, truth)
            bad = ~(e_got <= 2 * e_ref + floor)
            assert not bad[~easy].any(), (e_got[bad].max(), e_ref[bad].max(), cond[bad].max())

    def check_inv(got):
        check(got, ref_inv, true_inv)
        eye = np.einsum('bij,bjk->bik', a64, got.astype(np.float64))
        resid = np.abs(eye - np.eye(n)).reshape(nb, -1).max(-1)
        assert (resid <= 4 * floor * n).all(), resid.max()        # A A^-1 = I to n * eps * cond

    for rep in range(2):
        check_inv(B.batchinv(t(a, dev)).cpu().numpy())
        check(B.batchdet(t(a, dev)).cpu().numpy()[:, None], ref_det[:, None], true_det[:, None])
    at = t(a.transpose(0, 2, 1).copy(), dev).transpose(-1, -2)
    check_inv(B.batchinv(at).cpu().numpy())


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', range(9, 17))
def test_sym_invert_diag_large_orders(dev, oracle, dn, M):
    """`sym_invert(diag=True)` at orders 9..16 runs the one-matrix-per-16-lanes kernel
    (nfm_rowwave.hip) for both dtypes; ragged batches cover a partly filled last tile of 16."""
    dtype = np.float32 if dn == 'f32' else np.float64
    S = N().sym
    for n in (1, 15, 16, 17, 1000 + M):
        mat, _ = spd_np(n, M, dtype, 500 + M)
        got = S.sym_invert(t(mat, dev), diag=True).cpu().numpy()
        assert got.shape == (n, M) and relerr(got, oracle.sym_invert(mat, diag=True)) <= TOL[dn]


@pytest.mark.parametrize('M', [14, 15, 16])
def test_rowwave_ragged_tiles_f64(dev, oracle, M):
    """float64 orders 14..16 run one matrix per 16 lanes, 16 matrices per workgroup: batches that
    leave the last tile partly filled (and batches smaller than one tile), every op; the register
    kernels they replace needed scratch memory there (`private_segment_fixed_size` is now 0 for
    every kernel of nfm_rowwave.hip: tests/test_abi_host.py checks the code object)."""
    S, B = N().sym, N().batched
    for n in (1, 2, 15, 16, 17, 33, 255):
        mat, vec = spd_np(n, M, np.float64, 40 + M)
        assert relerr(S.sym_solve(t(mat, dev), t(vec, dev)).cpu().numpy(), oracle.sym_solve(mat, vec)) <= TOL['f64']
        assert relerr(S.sym_solve(t(mat, dev), t(vec, dev), eps=0.5).cpu().numpy(),
                      oracle.sym_solve(mat + np.r_[np.full(M, 0.5), np.zeros(M * (M - 1) // 2)], vec)) <= TOL['f64']
        assert relerr(S.sym_invert(t(mat, dev)).cpu().numpy(), oracle.sym_invert(mat)) <= TOL['f64']
        assert relerr(S.sym_det(t(mat, dev)).cpu().numpy(), oracle.sym_det(mat)) <= 4 * TOL['f64']
        rng = np.random.default_rng(n + M)
        a = rng.standard_normal((n, M, M)) + 8 * np.eye(M)
        a[::3, 0, 0] = 0
        assert relerr(B.batchinv(t(a, dev)).cpu().numpy(), oracle.batch_inv(a)) <= 1e-10
        d, do = B.batchdet(t(a, dev)).cpu().numpy(), oracle.batch_det(a)
        assert np.abs(d / do - 1).max() <= 1e-10           # every determinant, sign included
    # singular input: inf / NaN, no hang, no fault (the reference divides by the zero pivot too)
    z = np.zeros((5, M, M))
    assert not np.isfinite(B.batchinv(t(z, dev)).cpu().numpy()).any()
    assert (B.batchdet(t(z, dev)).cpu().numpy() == 0).all()


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_orders_above_16_take_the_references_route_on_device(dev, dn):
    """orders > 16 are outside the kernels (`NFM_MAX_DIM`); the facade then does what the reference
    does for every order > 4: densify and call torch.linalg on the device (`_impl/sym.py:392-396`,
    `_impl/batched.py:119-120`) -- no ValueError, no CPU round trip"""
    dtype = np.float32 if dn == 'f32' else np.float64
    tol = 2e-5 if dn == 'f32' else 1e-11
    M, n = 20, 37
    mat, vec = spd_np(n, M, dtype, 9)
    S, B = N().sym, N().batched
    full = S.sym_to_full(t(mat, dev))
    assert full.shape == (n, M, M) and torch.equal(full, full.transpose(-1, -2))
    f64 = full.double().cpu().numpy()
    x = S.sym_solve(t(mat, dev), t(vec, dev))
    assert x.is_cuda and relerr(x.cpu().numpy(), np.linalg.solve(f64, vec.astype(np.float64)[..., None])[..., 0]) <= tol
    assert relerr(S.sym_matvec(t(mat, dev), t(vec, dev)).cpu().numpy(), np.einsum('bij,bj->bi', f64, vec)) <= tol
    inv = S.sym_invert(t(mat, dev))
    assert inv.shape == mat.shape
    assert relerr(S.sym_to_full(inv).cpu().numpy(), np.linalg.inv(f64)) <= tol
    assert relerr(S.sym_invert(t(mat, dev), diag=True).cpu().numpy(), np.diagonal(np.linalg.inv(f64), axis1=1, axis2=2)) <= tol
    assert relerr(S.sym_det(t(mat, dev)).cpu().numpy(), np.linalg.det(f64)) <= 10 * tol
    a = t(f64.astype(dtype), dev)
    assert relerr(B.batchinv(a).cpu().numpy(), np.linalg.inv(f64)) <= tol
    assert relerr(B.batchdet(a).cpu().numpy(), np.linalg.det(f64)) <= 10 * tol


def test_orders_above_16_backward(dev):
    """orders > 16 with requires_grad: forward AND backward go through torch.linalg on the device (the custom
    autograd Functions end in kernels capped at order 16: round 2 raised only at `.backward()`); gradients
    against the dense formulas"""
    M, n = 20, 6
    mat_np, vec_np = spd_np(n, M, np.float64, 11)
    S = N().sym
    mat = t(mat_np, dev).requires_grad_(True)
    vec = t(vec_np, dev).requires_grad_(True)
    x = S.sym_solve(mat, vec)
    (x * x).sum().backward()
    assert mat.grad is not None and vec.grad is not None and mat.grad.shape == mat.shape
    # d/dv sum(x^2) = 2 A^-1 x
    full = S.sym_to_full(mat.detach())
    want_v = 2 * torch.linalg.solve(full, x.detach().unsqueeze(-1)).squeeze(-1)
    assert relerr(vec.grad.cpu().numpy(), want_v.cpu().numpy()) <= 1e-10
    for fn in (lambda m: S.sym_invert(m).sum(), lambda m: S.sym_det(m).sum(), lambda m: S.sym_matvec(m, vec.detach()).sum(),
               lambda m: S.sym_invert(m, diag=True).sum()):
        m2 = t(mat_np, dev).requires_grad_(True)
        fn(m2).backward()
        assert m2.grad is not None and torch.isfinite(m2.grad).all() and m2.grad.abs().sum() > 0


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', [9, 12, 13, 16])
def test_one_matrix_many_vectors_large_orders(dev, oracle, dn, M):
    """`mat` broadcast against a batch of vectors at orders 9..16 (`_impl/sym.py:371`: one Hessian, a field of
    gradients): the workgroup factors the matrix once and streams the vectors (sym_bcast_big_kernel) --
    sym_solve (with eps, full-matrix storage, strided / channel-first vectors, several slabs), sym_matvec /
    addmatvec / submatvec; against the oracle on the expanded operands"""
    dtype = np.float32 if dn == 'f32' else np.float64
    n = 4096 + 517
    mat, vec = spd_np(n, M, dtype, 1900 + M)
    one = mat[:1]                                           # (1, K)
    S = N().sym
    md, vd = t(one, dev), t(vec, dev)
    rep = np.broadcast_to(one, (n, one.shape[-1]))
    got = S.sym_solve(md, vd).cpu().numpy()
    assert got.shape == (n, M) and relerr(got, oracle.sym_solve(rep, vec)) <= TOL[dn]
    # the same call on materialised operands takes the per-record kernels: the two agree to the same bar
    assert relerr(got, S.sym_solve(t(rep.copy(), dev), vd).cpu().numpy()) <= TOL[dn]
    # eps, out=, channel-first vectors / outputs
    eps = [0.5] * M
    ref_eps = oracle.sym_solve(rep + np.concatenate([np.full(M, 0.5), np.zeros(one.shape[-1] - M)]).astype(dtype), vec)
    assert relerr(S.sym_solve(md, vd, eps=eps).cpu().numpy(), ref_eps) <= TOL[dn]
    vcf = vd.t().contiguous().t()
    ocf = torch.empty(M, n, dtype=vd.dtype, device=dev).t()
    S.sym_solve(md, vcf, out=ocf)
    assert relerr(ocf.cpu().numpy(), oracle.sym_solve(rep, vec)) <= TOL[dn]
    # full-matrix storage (NN = M^2) broadcast
    full = S.sym_to_full(md)                                # (1, M, M)
    assert relerr(S.sym_solve(full.reshape(1, M * M), vd).cpu().numpy(), oracle.sym_solve(rep, vec)) <= TOL[dn]
    # several slabs: (B, 1, K) against (B, n, M)
    B = 3
    mats, vecs = spd_np(B, M, dtype, 7)[0], vec[: B * 1500].reshape(B, 1500, M)
    gs = S.sym_solve(t(mats[:, None], dev), t(vecs, dev)).cpu().numpy()
    for b in range(B):
        assert relerr(gs[b], oracle.sym_solve(np.broadcast_to(mats[b], (1500, mats.shape[-1])), vecs[b])) <= TOL[dn]
    # matvec family: reference fma chains per row -> agreement to rounding of an M-term sum
    mv = oracle.sym_matvec(rep, vec)
    tolmv = 4 * M * EPS[dn]
    assert relerr(S.sym_matvec(md, vd).cpu().numpy(), mv) <= tolmv
    inp = np.random.default_rng(5).standard_normal((n, M)).astype(dtype)
    assert relerr(S.sym_addmatvec(t(inp, dev), md, vd).cpu().numpy(), inp + mv) <= tolmv
    assert relerr(S.sym_submatvec(t(inp, dev), md, vd).cpu().numpy(), inp - mv) <= tolmv
    # singular matrix: inf / nan like a division, no hang
    z = torch.zeros(1, one.shape[-1], dtype=vd.dtype, device=dev)
    assert not torch.isfinite(S.sym_solve(z, vd)).any()


def sym_indefinite_np(n, M, dtype, seed, every):
    """a positive definite batch in which every `every`-th matrix is made indefinite (a well
    conditioned one: eigenvalues +-[1, 2], random orthogonal basis), compact storage"""
    rng = np.random.default_rng(seed)
    mat, vec = spd_np(n, M, np.float64, seed)
    idx = np.arange(0, n, every)
    Q, _ = np.linalg.qr(rng.standard_normal((len(idx), M, M)))
    lam = rng.uniform(1, 2, (len(idx), M)) * np.where(rng.random((len(idx), M)) < 0.5, -1.0, 1.0)
    lam[:, 0], lam[:, 1] = -np.abs(lam[:, 0]), np.abs(lam[:, 1])          # never definite
    A = np.einsum('nij,nj,nkj->nik', Q, lam, Q)
    iu = [(i, j) for i in range(M) for j in range(i + 1, M)]
    c = np.concatenate([np.stack([A[:, i, i] for i in range(M)], -1), np.stack([A[:, i, j] for i, j in iu], -1)], -1)
    mat[idx] = c
    return mat.astype(dtype), vec.astype(dtype), idx


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', range(9, 17))
def test_sym_large_orders_not_positive_definite(dev, oracle, dn, M):
    """orders 9..16 try the unpivoted factorisation of positive definite matrices first (nfm_spd.hip) and
    redo a wavefront with the pivoted elimination when ONE of its 64 matrices is not: batches with an
    indefinite matrix in every wavefront, in some wavefronts, and a zero / a NaN matrix -- same answers as
    the reference's LU (`_impl/sym.py:392-396`) for all of them, neighbours of the odd ones included"""
    dtype = np.float32 if dn == 'f32' else np.float64
    S = N().sym
    for n, every in ((1000 + M, 7), (1000 + M, 300), (3, 2), (64, 64), (65, 64)):
        mat, vec, idx = sym_indefinite_np(n, M, dtype, 70 + M + every, every)
        assert relerr(S.sym_solve(t(mat, dev), t(vec, dev)).cpu().numpy(), oracle.sym_solve(mat, vec)) <= 4 * TOL[dn]
        assert relerr(S.sym_invert(t(mat, dev)).cpu().numpy(), oracle.sym_invert(mat)) <= 4 * TOL[dn]
        assert relerr(S.sym_invert(t(mat, dev), diag=True).cpu().numpy(), oracle.sym_invert(mat, diag=True)) <= 4 * TOL[dn]
        d, do = S.sym_det(t(mat, dev)).cpu().numpy().astype(np.float64), oracle.sym_det(mat).astype(np.float64)
        assert np.abs(d / do - 1).max() <= 16 * TOL[dn]                  # every determinant, sign included
        # in place: the wavefront that falls back has stored nothing before it does
        v2 = t(vec, dev)
        S.sym_solve_(t(mat, dev), v2)
        assert relerr(v2.cpu().numpy(), oracle.sym_solve(mat, vec)) <= 4 * TOL[dn]
        m2 = t(mat, dev)
        S.sym_invert_(m2)
        assert relerr(m2.cpu().numpy(), oracle.sym_invert(mat)) <= 4 * TOL[dn]
        # eps on the diagonal goes through both paths
        e = 0.25
        assert relerr(S.sym_solve(t(mat, dev), t(vec, dev), eps=e).cpu().numpy(),
                      oracle.sym_solve(mat + np.r_[np.full(M, e), np.zeros(M * (M - 1) // 2)].astype(dtype), vec)) <= 16 * TOL[dn]
    # a singular and a NaN matrix among definite ones: inf / NaN for them, the neighbours untouched
    mat, vec = spd_np(130, M, dtype, 5 + M)
    ref = oracle.sym_solve(mat, vec)
    mat[17] = 0
    mat[99, 3] = np.nan
    got = S.sym_solve(t(mat, dev), t(vec, dev)).cpu().numpy()
    keep = np.ones(130, bool)
    keep[[17, 99]] = False
    assert relerr(got[keep], ref[keep]) <= TOL[dn] and not np.isfinite(got[17]).all() and np.isnan(got[99]).any()


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', range(9, 17))
def test_general_large_orders_diagonal_pivots_first(dev, oracle, dn, n):
    """`batchinv` / `batchdet` at orders 9..16 try the elimination WITHOUT row exchanges first (the diagonal is
    accepted while it is within a factor 8 of its column's maximum: nfm_spd.hip) and redo a wavefront with the
    pivoted elimination when one of its 64 matrices needed an exchange.  `test_general_large_orders` has such a
    matrix in every wavefront; here: none at all (the no-exchange path alone), one every 300 (both paths in one
    launch), a permuted identity-like batch (every matrix needs exchanges), ragged sizes -- against the oracle's
    pivoted LU (`_impl/batched.py:119-120`, `:53-54`) at TOL, and A A^-1 = I."""
    dtype = np.float32 if dn == 'f32' else np.float64
    B = N().batched
    rng = np.random.default_rng(170 + n)
    for nb, every in ((1500 + n, 0), (1500 + n, 300), (1, 0), (63, 0), (65, 64)):
        a = (rng.standard_normal((nb, n, n)) + 8 * np.eye(n)).astype(dtype)
        if every:
            a[::every] = a[::every][:, ::-1]          # rows reversed: the diagonal is the wrong pivot everywhere
        inv = B.batchinv(t(a, dev)).cpu().numpy()
        assert relerr(inv, oracle.batch_inv(a)) <= 4 * TOL[dn]
        eye = np.einsum('bij,bjk->bik', a.astype(np.float64), inv.astype(np.float64))
        assert np.abs(eye - np.eye(n)).max() <= 64 * n * EPS[dn]
        d, do = B.batchdet(t(a, dev)).cpu().numpy().astype(np.float64), oracle.batch_det(a).astype(np.float64)
        assert np.abs(d / do - 1).max() <= 16 * TOL[dn]              # every determinant, sign included
    # a diagonal that is acceptable at first and not later (the test is made at every step)
    a = (rng.standard_normal((200, n, n)) * 0.1 + np.eye(n)).astype(dtype)
    a[::3, n - 1, n - 1] = 0
    a[::3, n - 1, n - 2] = 1
    assert relerr(B.batchinv(t(a, dev)).cpu().numpy(), oracle.batch_inv(a)) <= 64 * TOL[dn]
    # singular and NaN matrices among good ones: inf / NaN for them only
    a = (rng.standard_normal((130, n, n)) + 8 * np.eye(n)).astype(dtype)
    ref = oracle.batch_inv(a)
    a[17] = 0
    a[99, 3, 3] = np.nan
    got = B.batchinv(t(a, dev)).cpu().numpy()
    keep = np.ones(130, bool)
    keep[[17, 99]] = False
    assert relerr(got[keep], ref[keep]) <= TOL[dn] and not np.isfinite(got[17]).all() and np.isnan(got[99]).any()
    assert B.batchdet(t(a, dev)).cpu().numpy()[17] == 0


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', [9, 12, 13, 16])
def test_sym_large_orders_any_strides(dev, oracle, dn, M):
    """orders 9..16 with channel-first fields, two batch levels, padded / interleaved records and a broadcast
    vector: the strided positive-definite-first kernel (`spd_strided_kernel`, nfm_spd.hip) -- every lane addresses
    its own record element by element, the fallback gathers / scatters its group -- on batches with indefinite
    matrices mixed in, all four ops, out= buffers of every layout, in place."""
    dtype = np.float32 if dn == 'f32' else np.float64
    K = M * (M + 1) // 2
    S = N().sym
    B, X, Y = 2, 19, 11
    n = B * X * Y
    tol = 4 * TOL[dn]
    for every in (0, 29):
        if every:
            mat, vec, _ = sym_indefinite_np(n, M, dtype, 300 + M, every)
        else:
            mat, vec = spd_np(n, M, dtype, 300 + M)
        ref, refi = oracle.sym_solve(mat, vec), oracle.sym_invert(mat)
        refd, refdet = oracle.sym_invert(mat, diag=True), oracle.sym_det(mat).astype(np.float64)
        mat4, vec4 = t(mat, dev).reshape(B, X, Y, K), t(vec, dev).reshape(B, X, Y, M)
        # (B, C, X, Y) channel-first storage viewed channel-last: no copy, two batch levels
        mat_cf = mat4.movedim(-1, 1).contiguous().movedim(1, -1)
        vec_cf = vec4.movedim(-1, 1).contiguous().movedim(1, -1)
        assert not mat_cf.is_contiguous()
        assert relerr(S.sym_solve(mat_cf, vec_cf).cpu().numpy().reshape(n, M), ref) <= tol
        assert relerr(S.sym_solve(mat_cf, vec4).cpu().numpy().reshape(n, M), ref) <= tol          # mixed layouts
        assert relerr(S.sym_invert(mat_cf).cpu().numpy().reshape(n, K), refi) <= tol
        assert relerr(S.sym_invert(mat_cf, diag=True).cpu().numpy().reshape(n, M), refd) <= tol
        d = S.sym_det(mat_cf).cpu().numpy().astype(np.float64).reshape(n)
        assert np.abs(d / refdet - 1).max() <= 16 * TOL[dn]
        # sym_matvec / sym_addmatvec / sym_submatvec on the same storage: bit-identical to the oracle in any layout
        refmv = oracle.sym_matvec(mat, vec)
        mv = S.sym_matvec(mat_cf, vec_cf)
        assert mv.stride() == vec_cf.stride() and np.array_equal(mv.cpu().numpy().reshape(n, M), refmv)
        assert np.array_equal(S.sym_matvec(mat_cf, vec4).cpu().numpy().reshape(n, M), refmv)
        assert np.array_equal(S.sym_addmatvec(vec_cf, mat_cf, vec_cf).cpu().numpy().reshape(n, M), vec + refmv)
        assert np.array_equal(S.sym_submatvec(vec4, mat_cf, vec_cf).cpu().numpy().reshape(n, M), vec - refmv)
        # channel-first output buffers through out=
        out_cf = torch.empty(B, M, X, Y, dtype=vec4.dtype, device=dev).movedim(1, -1)
        r = S.sym_solve(mat_cf, vec_cf, out=out_cf)
        assert r.data_ptr() == out_cf.data_ptr() and relerr(out_cf.cpu().numpy().reshape(n, M), ref) <= tol
        inv_cf = torch.empty(B, K, X, Y, dtype=vec4.dtype, device=dev).movedim(1, -1)
        S.sym_invert(mat_cf, out=inv_cf)
        assert relerr(inv_cf.cpu().numpy().reshape(n, K), refi) <= tol
        # pure SoA: (K, n).T; in place on channel-first storage
        mat_soa, vec_soa = t(mat, dev).t().contiguous().t(), t(vec, dev).t().contiguous().t()
        assert relerr(S.sym_solve(mat_soa, vec_soa).cpu().numpy(), ref) <= tol
        v2 = vec_soa.clone(memory_format=torch.preserve_format)
        S.sym_solve_(mat_soa, v2)
        assert relerr(v2.cpu().numpy(), ref) <= tol
        m2 = mat_soa.clone(memory_format=torch.preserve_format)
        S.sym_invert_(m2)
        assert relerr(m2.cpu().numpy(), refi) <= tol
        # every other record, padded records, one vector for every matrix
        assert relerr(S.sym_solve(t(mat, dev)[::2], t(vec, dev)[::2]).cpu().numpy(), ref[::2]) <= tol
        pad = torch.zeros(n, K + 3, dtype=mat4.dtype, device=dev)
        pad[:, :K] = t(mat, dev)
        assert relerr(S.sym_solve(pad[:, :K], t(vec, dev)).cpu().numpy(), ref) <= tol
        assert relerr(S.sym_invert(pad[:, :K]).cpu().numpy(), refi) <= tol
        ref_b = oracle.sym_solve(mat, np.broadcast_to(vec[:1], (n, M)))
        assert relerr(S.sym_solve(t(mat, dev), t(vec[0], dev)).cpu().numpy(), ref_b) <= tol
        # eps through the strided kernel
        e = 0.25
        assert relerr(S.sym_solve(mat_cf, vec_cf, eps=e).cpu().numpy().reshape(n, M),
                      oracle.sym_solve(mat + np.r_[np.full(M, e), np.zeros(K - M)].astype(dtype), vec)) <= 4 * tol


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', [9, 12, 16])
def test_pivoting_always_skips_the_positive_definite_attempt(dev, oracle, dn, M):
    """`pivoting='always'` (keyword-only extension, `sym.PIVOTING`; NFM_MAT_PIVOTED / NFM_INVERT_PIVOTED in the C ABI):
    orders 9..16 go straight to the pivoted elimination -- for callers whose matrices are indefinite.  Same answers as
    'auto' within the tolerance, on definite and indefinite batches, contiguous and channel-first operands."""
    dtype = np.float32 if dn == 'f32' else np.float64
    S = N().sym
    n = 700 + M
    for every in (0, 1):
        if every:
            mat, vec, _ = sym_indefinite_np(n, M, dtype, 900 + M, 1)      # every matrix indefinite
        else:
            mat, vec = spd_np(n, M, dtype, 900 + M)
        ref, refi = oracle.sym_solve(mat, vec), oracle.sym_invert(mat)
        for m_, v_ in ((t(mat, dev), t(vec, dev)), (t(mat, dev).t().contiguous().t(), t(vec, dev).t().contiguous().t())):
            for mode in ('auto', 'always'):
                assert relerr(S.sym_solve(m_, v_, pivoting=mode).cpu().numpy(), ref) <= 4 * TOL[dn]
                assert relerr(S.sym_invert(m_, pivoting=mode).cpu().numpy(), refi) <= 4 * TOL[dn]
                assert relerr(S.sym_invert(m_, diag=True, pivoting=mode).cpu().numpy(),
                              oracle.sym_invert(mat, diag=True)) <= 4 * TOL[dn]
    with pytest.raises(ValueError):
        S.sym_solve(t(mat, dev), t(vec, dev), pivoting='never')


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', [9, 16])
def test_large_batches_redo_marked_groups_in_a_second_launch(dev, oracle, dn, M):
    """batches of >= 2^20 matrices whose output aliases no input: the groups of 16 that hold a matrix which fails the
    no-exchange test are MARKED by the first launch and redone by `redo_kernel` (nfm_spd.hip); smaller batches and
    in-place calls redo them inside the first kernel (every other test of this file).  An indefinite (general: a
    row-reversed) matrix every 4099, checked on the marked groups, their neighbours and both ends of the batch."""
    dtype = np.float32 if dn == 'f32' else np.float64
    S, B = N().sym, N().batched
    n = (1 << 20) + 77
    every = 4099
    bad = np.arange(5, n, every)
    sel = np.unique(np.concatenate([np.arange(0, 600), np.arange(n - 300, n)] +
                                   [np.arange(max(b // 16 * 16 - 16, 0), min(b // 16 * 16 + 32, n)) for b in bad[:40]]))
    # compact symmetric
    mat, vec = spd_np(n, M, dtype, 4000 + M)
    mi, _, _ = sym_indefinite_np(len(bad), M, dtype, 4100 + M, 1)
    mat[bad] = mi
    ms, vs = mat[sel], vec[sel]
    assert relerr(S.sym_solve(t(mat, dev), t(vec, dev)).cpu().numpy()[sel], oracle.sym_solve(ms, vs)) <= 4 * TOL[dn]
    assert relerr(S.sym_invert(t(mat, dev)).cpu().numpy()[sel], oracle.sym_invert(ms)) <= 4 * TOL[dn]
    assert relerr(S.sym_invert(t(mat, dev), diag=True).cpu().numpy()[sel], oracle.sym_invert(ms, diag=True)) <= 4 * TOL[dn]
    d = S.sym_det(t(mat, dev)).cpu().numpy().astype(np.float64)[sel]
    assert np.abs(d / oracle.sym_det(ms).astype(np.float64) - 1).max() <= 16 * TOL[dn]
    # in place at the same size: the in-kernel path
    v2 = t(vec, dev)
    S.sym_solve_(t(mat, dev), v2)
    assert relerr(v2.cpu().numpy()[sel], oracle.sym_solve(ms, vs)) <= 4 * TOL[dn]
    del mat, vec, v2
    if dn == 'f64' and M > 11:
        return
    # general
    rng = np.random.default_rng(4200 + M)
    a = (rng.standard_normal((n, M, M)) + 8 * np.eye(M)).astype(dtype)
    a[bad] = a[bad][:, ::-1]
    asel = a[sel]
    assert relerr(B.batchinv(t(a, dev)).cpu().numpy()[sel], oracle.batch_inv(asel)) <= 4 * TOL[dn]
    d = B.batchdet(t(a, dev)).cpu().numpy().astype(np.float64)[sel]
    assert np.abs(d / oracle.batch_det(asel).astype(np.float64) - 1).max() <= 16 * TOL[dn]
