"""Pin the CPU oracle to golden vectors produced by the real reference
(tests/golden/make_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest
from conftest import TOL, relerr

MS = (1, 2, 3, 4, 5, 6, 7, 8, 12, 16)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', MS)
def test_sym_family(oracle, golden_sym, dn, M):
    g, k = golden_sym, f'{dn}_M{M}_'
    mat, vec, inp = g[k + 'mat'], g[k + 'vec'], g[k + 'inp']
    got = {
        'solve': oracle.sym_solve(mat, vec),
        'matvec': oracle.sym_matvec(mat, vec),
        'addmatvec': oracle.sym_matvec(mat, vec, inp, +1),
        'submatvec': oracle.sym_matvec(mat, vec, inp, -1),
        'invert': oracle.sym_invert(mat),
        'invert_diag': oracle.sym_invert(mat, diag=True),
        'det': oracle.sym_det(mat),
        'to_full': oracle.sym_to_full(mat),
        'outer': oracle.sym_outer(vec),
    }
    exact = {'matvec', 'addmatvec', 'submatvec', 'to_full', 'outer'}
    for name, val in got.items():
        ref = g[k + name]
        assert val.shape == ref.shape and val.dtype == ref.dtype
        if M <= 4 or name in exact:
            # closed forms follow the reference's operation order: bit-identical
            assert np.array_equal(val, ref), (name, relerr(val, ref))
        else:
            # M > 4: the reference calls LAPACK LU; same algorithm, different summation order
            assert relerr(val, ref) <= TOL[dn], name


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', MS[1:])
def test_sym_indefinite(oracle, golden_sym, dn, M):
    g, k = golden_sym, f'{dn}_M{M}_'
    s = oracle.sym_solve(g[k + 'mat_indef'], g[k + 'vec'])
    i = oracle.sym_invert(g[k + 'mat_indef'])
    if M <= 4:
        assert np.array_equal(s, g[k + 'solve_indef']) and np.array_equal(i, g[k + 'invert_indef'])
    else:
        assert relerr(s, g[k + 'solve_indef']) <= TOL[dn]
        assert relerr(i, g[k + 'invert_indef']) <= TOL[dn]


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', MS)
def test_sym_kinds(oracle, golden_sym, dn, M):
    g, k = golden_sym, f'{dn}_M{M}_'
    mat, vec = g[k + 'mat'], g[k + 'vec']
    if M > 1:   # for M == 1 every kind coincides with the compact reading
        assert np.array_equal(oracle.sym_solve(mat[:, :M], vec), g[k + 'solve_diag'])
        assert np.array_equal(oracle.sym_matvec(mat[:, :M], vec), g[k + 'matvec_diag'])
        assert np.array_equal(oracle.sym_solve(mat[:, :1], vec), g[k + 'solve_scal'])
        assert np.array_equal(oracle.sym_matvec(mat[:, :1], vec), g[k + 'matvec_scal'])
    if M > 2:
        full = g[k + 'to_full'].reshape(len(mat), M * M)
        assert relerr(oracle.sym_solve(full, vec), g[k + 'solve']) <= TOL[dn]
        assert relerr(oracle.sym_matvec(full, vec), g[k + 'matvec']) <= TOL[dn]


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('kd', [(1, 1), (2, 2), (3, 3), (3, 2), (4, 4), (2, 3)])
def test_sym_matmul(oracle, golden_sym, dn, kd):
    g, k = golden_sym, f'{dn}_k{kd[0]}_d{kd[1]}_'
    assert np.array_equal(oracle.sym_matmul(g[k + 'j'], g[k + 'h']), g[k + 'matmul'])


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', MS)
def test_batched(oracle, golden_batched, dn, n):
    g, k = golden_batched, f'{dn}_n{n}_'
    a, v = g[k + 'a'], g[k + 'v']
    assert relerr(oracle.batch_inv(a), g[k + 'inv']) <= TOL[dn]
    assert relerr(oracle.batch_det(a), g[k + 'det']) <= TOL[dn]
    assert relerr(oracle.batch_matvec(a, v), g[k + 'matvec']) <= TOL[dn]
    if n in (2, 3):  # the TorchScript closed forms, incl. the det perturbation: bit-identical
        assert np.array_equal(oracle.batch_inv(a, closed=True), g[k + 'inv_ts'])
        assert np.array_equal(oracle.batch_det(a, closed=True), g[k + 'det_ts'])


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_batched_rect(oracle, golden_batched, dn):
    g = golden_batched
    assert relerr(oracle.batch_matvec(g[dn + '_rect_a'], g[dn + '_rect_v']), g[dn + '_rect_matvec']) <= TOL[dn]


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('size', [1, 63, 64, 65, 4097, 20011])
@pytest.mark.parametrize('nn', ['nan0', 'nan1', 'nanall'])
def test_reduce_full(oracle, golden_reduce, dn, size, nn):
    g, k = golden_reduce, f'{dn}_{size}_{nn}_'
    x = g[k + 'x']
    fin = x[np.isfinite(x)]
    scale = max(float(np.abs(fin).sum()), 1e-30)
    for op in ('nansum', 'sum'):
        r, e = oracle.reduce(op, x), g[k + op]
        assert (np.isnan(r) and np.isnan(e)) or r == e or abs(float(r) - float(e)) <= TOL[dn] * scale
    r, e = oracle.reduce('nansum', x, out_f64=True), g[k + 'nansum64']
    assert (np.isnan(r) and np.isnan(e)) or r == e or abs(float(r) - float(e)) <= 1e-12 * scale
    for op in ('nanmax', 'nanmin', 'max', 'min'):
        r, e = oracle.reduce(op, x), g[k + op]
        assert (np.isnan(r) and np.isnan(e)) or r == e, op


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_reduce_dims(oracle, golden_reduce, dn):
    g = golden_reduce
    x = g[dn + '_nd_x']
    for dim, name in ((0, 'd0'), (1, 'd1'), (2, 'd2'), (-1, 'dm1'), ((0, 2), 'd02'), ((1, 2), 'd12')):
        r = oracle.reduce('nansum', x, dim)
        e = g[f'{dn}_nd_nansum_{name}']
        assert r.shape == e.shape and relerr(r, e) <= TOL[dn]
        rk = oracle.reduce('nansum', x, dim, keepdim=True)
        assert rk.shape == g[f'{dn}_nd_nansum_keep_{name}'].shape
        s = oracle.reduce('sum', x, dim)
        es = g[f'{dn}_nd_sum_{name}']
        assert np.array_equal(np.isnan(s), np.isnan(es))
        assert relerr(np.nan_to_num(s), np.nan_to_num(es)) <= TOL[dn]


def test_reduce_empty(oracle, golden_reduce):
    assert oracle.reduce('nansum', np.zeros(0, np.float32)) == golden_reduce['f32_empty_nansum'] == 0


# ------------------------------------------------------------------ QR family
# f64 and orders <= 5 (where the reference itself runs): TOL against the golden vectors.
# float32 beyond: the error model of conftest (`parity_ok`), the truth being the float64 oracle
# on the same inputs -- the reference's own fp32 Householder reduction is 9e-6 off at n = 12.
QR_NS = (1, 2, 3, 4, 5, 6, 8, 12)


@pytest.fixture(scope='session')
def golden_qr():
    import os
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, 'qr.npz'))


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_qr_givens(oracle, golden_qr, dn):
    g = golden_qr
    c, s = oracle.givens(g[dn + '_givens_x'], g[dn + '_givens_y'])
    assert relerr(c, g[dn + '_givens_c']) <= TOL[dn] and relerr(s, g[dn + '_givens_s']) <= TOL[dn]
    assert c[0] == 1 and s[0] == 0          # x = y = 0 -> identity rotation


def qr_family_cases(O, g, k, n, f64=False):
    """every (name, value) the QR-family goldens of order n hold, computed by backend `O` from the
    golden inputs (cast to float64 when f64: that run is the truth of the error model)"""
    up = (lambda x: x.astype(np.float64)) if f64 else (lambda x: x)
    a, v, hz = up(g[k + 'a']), up(g[k + 'hh_x']), up(g[k + 'hz'])
    out = {}
    for b in sorted({0, n - 1}):
        u, al = O.householder(v, b)
        out[f'hh_u_b{b}'], out[f'hh_alpha_b{b}'] = u, al
    # the reflector fed to the apply cases is the golden one, so that only the apply is compared
    u = up(g[k + 'hh_u_b0'])
    for side in ('left', 'right', 'both'):
        out[f'hh_apply_{side}'] = O.householder_apply(a, u, side)
    if n >= 3:
        u2, _ = O.householder(up(g[k + 'hh_x'])[:, 1:])
        out['hh_apply_short'] = O.householder_apply(a, u2, 'both')
        out['hh_apply_two_inv'] = O.householder_apply(a, [u, u2], 'left', True)
    if n >= 2:
        cc, ss = up(g[k + 'ga_c'])[:, None], up(g[k + 'ga_s'])[:, None]
        for side in ('left', 'right', 'both'):
            out[f'givens_apply_{side}'] = O.givens_apply(a, cc, ss, 0, n - 1, side)
        out['givens_apply_default_j'] = O.givens_apply(a, cc, ss, 0, None, 'left')
    h, us = O.hessenberg(a, True)
    out['hess'] = h
    for i, ui in enumerate(us):
        out[f'hess_u{i}'] = ui
    out['qrh_q'], out['qrh_r'] = O.qr_hessenberg(hz)
    out['rq_true'] = O.rq_hessenberg(hz)
    return out


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', QR_NS)
def test_qr_family(oracle, golden_qr, dn, n):
    from conftest import parity_ok, within_model
    g, k = golden_qr, f'{dn}_n{n}_'
    got = qr_family_cases(oracle, g, k, n)
    truth = qr_family_cases(oracle, g, k, n, f64=True) if dn == 'f32' else {}
    for name, val in got.items():
        assert parity_ok(val, g[k + name], n, dn, truth.get(name)), (name, relerr(val, g[k + name]))
    assert not np.isnan(got['hh_u_b0']).any()
    if n <= 3:
        assert relerr(oracle.rq_hessenberg(g[k + 'hz'], true_rq=False), g[k + 'rq_ref']) <= TOL[dn]
    # eigenvalues of the symmetrised matrix: the golden vector is LAPACK in float64 = the truth
    ev = np.sort(oracle.eig_sym(g[k + 'sym']), -1)
    assert within_model(ev, g[k + 'eigvalsh'], g[k + 'eigvalsh'], n, dn), relerr(ev, g[k + 'eigvalsh'])


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', [1, 2, 3, 4, 5])
def test_qr_sym_family(oracle, golden_qr, dn, n):
    """orders where the reference itself runs (quirk Q7): TOL, no scaling"""
    g, k, tol = golden_qr, f'{dn}_n{n}_', TOL[dn]
    a, sym = g[k + 'a'], g[k + 'sym']
    for up in (1, 0):
        t, us = oracle.hessenberg_sym(sym, bool(up), True, True)
        assert relerr(t, g[k + f'hess_sym_{up}']) <= tol
        for i, ui in enumerate(us):
            assert relerr(ui, g[k + f'hess_sym_{up}_u{i}']) <= tol
        # un-symmetrised input: only the requested triangle may be read
        assert relerr(oracle.hessenberg_sym(a, bool(up), True), g[k + f'hess_nonsym_{up}']) <= tol
        assert relerr(oracle.eig_sym(a, upper=bool(up)), g[k + f'eig_{up}']) <= tol
    # same ORDER (deflation order) and same eigenvector signs as the reference on one matrix
    assert relerr(oracle.eig_sym(sym), g[k + 'eig']) <= tol
    ev, evec = oracle.eig_sym(sym, True)
    assert relerr(ev, g[k + 'eig_u_val']) <= tol and relerr(evec, g[k + 'eig_u_vec']) <= 2 * tol
    # batched upstream call: same multiset (its order depends on the batch, quirk Q9)
    assert relerr(np.sort(ev, -1), np.sort(g[k + 'eig_batched'], -1)) <= tol
    assert relerr(oracle.rq_hessenberg(g[k + 'tri']), g[k + 'rq_tri']) <= tol
    eye = np.broadcast_to(np.eye(n, dtype=a.dtype), a.shape)
    assert relerr(oracle.rq_hessenberg(g[k + 'tri'], eye)[1], g[k + 'rq_tri_u']) <= tol
