"""GPU parity tests for the QR family (givens, householder, hessenberg, qr/rq_hessenberg,
eig_sym) through the C ABI: golden fixtures from the reference, the CPU oracle on seeded
inputs for every order 1..16 (incl. the orders where upstream raises), and properties."""
import os
import numpy as np
import pytest
import torch
from conftest import TOL, relerr, GOLDEN

pytestmark = pytest.mark.gpu
QR_NS = (1, 2, 3, 4, 5, 6, 8, 12)


def Q():
    import nitorch_fastmath_amd as N_
    return N_.qr


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def n_(x):
    return x.cpu().numpy()


def qr_tol(dn, n):
    return TOL[dn] * max(1.0, n * n / 4.0)


@pytest.fixture(scope='module')
def golden_qr():
    return np.load(os.path.join(GOLDEN, 'qr.npz'))


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_golden_givens(dev, golden_qr, dn):
    g = golden_qr
    c, s = Q().givens(t(g[dn + '_givens_x'], dev), t(g[dn + '_givens_y'], dev))
    assert relerr(n_(c), g[dn + '_givens_c']) <= TOL[dn] and relerr(n_(s), g[dn + '_givens_s']) <= TOL[dn]
    assert float(c[0]) == 1 and float(s[0]) == 0


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', QR_NS)
def test_golden_family(dev, golden_qr, dn, n):
    g, k, tol = golden_qr, f'{dn}_n{n}_', qr_tol(dn, n)
    a, v = t(g[k + 'a'], dev), t(g[k + 'hh_x'], dev)
    for b in sorted({0, n - 1}):
        u, al = Q().householder(v, basis=b, return_alpha=True)
        assert relerr(n_(u), g[k + f'hh_u_b{b}']) <= tol and relerr(n_(al), g[k + f'hh_alpha_b{b}']) <= tol
        assert not torch.isnan(u).any()
    u = Q().householder(v)
    for side in ('left', 'right', 'both'):
        assert relerr(n_(Q().householder_apply(a, u, side=side)), g[k + f'hh_apply_{side}']) <= tol
    if n >= 3:
        u2 = Q().householder(v[:, 1:])
        assert relerr(n_(Q().householder_apply(a, u2, side='both')), g[k + 'hh_apply_short']) <= tol
        r = Q().householder_apply(a, [u, u2], side='left', inverse=True)
        assert relerr(n_(r), g[k + 'hh_apply_two_inv']) <= tol
    if n >= 2:
        cc, ss = t(g[k + 'ga_c'], dev)[:, None], t(g[k + 'ga_s'], dev)[:, None]
        for side in ('left', 'right', 'both'):
            r = Q().givens_apply(a, cc, ss, 0, n - 1, side=side)
            assert relerr(n_(r), g[k + f'givens_apply_{side}']) <= tol
        assert relerr(n_(Q().givens_apply(a, cc, ss, 0, side='left')), g[k + 'givens_apply_default_j']) <= tol
    h, us = Q().hessenberg(a, compute_u=True)
    assert relerr(n_(h), g[k + 'hess']) <= tol
    assert len(us) == max(n - 2, 0)
    for i, ui in enumerate(us):
        assert relerr(n_(ui), g[k + f'hess_u{i}']) <= tol
    assert relerr(n_(Q().hessenberg(a)), g[k + 'hess']) <= tol
    q, r = Q().qr_hessenberg(t(g[k + 'hz'], dev))
    assert relerr(n_(q), g[k + 'qrh_q']) <= tol and relerr(n_(r), g[k + 'qrh_r']) <= tol
    assert relerr(n_(Q().rq_hessenberg(t(g[k + 'hz'], dev))), g[k + 'rq_true']) <= tol
    ev = np.sort(n_(Q().eig_sym(t(g[k + 'sym'], dev))), -1)
    assert relerr(ev, g[k + 'eigvalsh']) <= 4 * tol


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', [1, 2, 3, 4, 5])
def test_golden_sym_family(dev, golden_qr, dn, n):
    g, k, tol = golden_qr, f'{dn}_n{n}_', qr_tol(dn, n)
    a, sym = t(g[k + 'a'], dev), t(g[k + 'sym'], dev)
    for up in (1, 0):
        tt, us = Q().hessenberg_sym(sym, upper=bool(up), compute_u=True)
        assert relerr(n_(tt), g[k + f'hess_sym_{up}']) <= tol
        for i, ui in enumerate(us):
            assert relerr(n_(ui), g[k + f'hess_sym_{up}_u{i}']) <= tol
        assert relerr(n_(Q().hessenberg_sym(a, upper=bool(up))), g[k + f'hess_nonsym_{up}']) <= tol
        assert relerr(n_(Q().eig_sym(a, upper=bool(up))), g[k + f'eig_{up}']) <= 4 * tol
    assert relerr(n_(Q().eig_sym(sym)), g[k + 'eig']) <= 4 * tol        # same deflation ORDER
    ev, evec = Q().eig_sym(sym, compute_u=True)
    assert relerr(n_(ev), g[k + 'eig_u_val']) <= 4 * tol and relerr(n_(evec), g[k + 'eig_u_vec']) <= 8 * tol
    assert relerr(np.sort(n_(ev), -1), np.sort(g[k + 'eig_batched'], -1)) <= 4 * tol
    assert relerr(n_(Q().rq_hessenberg(t(g[k + 'tri'], dev))), g[k + 'rq_tri']) <= tol
    eye = torch.eye(n, dtype=a.dtype, device=dev).expand(a.shape)
    h2, u2 = Q().rq_hessenberg(t(g[k + 'tri'], dev), eye)
    assert relerr(n_(u2), g[k + 'rq_tri_u']) <= tol and relerr(n_(h2), g[k + 'rq_tri']) <= tol


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16])
def test_vs_oracle(dev, oracle, dn, n):
    """seeded inputs, every order (orders > 5 raise upstream), ragged batch"""
    dtype = np.float32 if dn == 'f32' else np.float64
    tol = qr_tol(dn, n)
    nb = 777 if n <= 8 else 130
    rng = np.random.default_rng(1000 + n)
    a = rng.standard_normal((nb, n, n)).astype(dtype)
    sym = ((a + a.transpose(0, 2, 1)) / 2).astype(dtype)
    ad, sd = t(a, dev), t(sym, dev)
    ev = n_(Q().eig_sym(sd))
    ref = oracle.eig_sym(sym)
    # identical algorithm, identical operation order: same deflation order; tolerance
    # covers the few matrices where one extra iteration is taken
    assert relerr(np.sort(ev, -1), np.sort(ref, -1)) <= 4 * tol
    assert relerr(np.sort(ev, -1), np.linalg.eigvalsh(sym.astype(np.float64))) <= 8 * tol
    frac_same_order = np.mean(np.abs(ev - ref).max(-1) <= 8 * tol * max(np.abs(ref).max(), 1))
    assert frac_same_order >= 0.98, frac_same_order
    ev2, evec = Q().eig_sym(sd, compute_u=True)
    ev2, evec = n_(ev2).astype(np.float64), n_(evec).astype(np.float64)
    s64 = sym.astype(np.float64)
    # A U = U diag(s), U orthonormal
    res = np.einsum('bij,bjk->bik', s64, evec) - evec * ev2[:, None, :]
    assert np.abs(res).max() <= 40 * tol * max(np.abs(s64).max(), 1)
    gram = np.einsum('bji,bjk->bik', evec, evec)
    assert np.abs(gram - np.eye(n)).max() <= 40 * tol
    for up in (True, False):
        assert relerr(n_(Q().hessenberg_sym(ad, upper=up)), oracle.hessenberg_sym(a, up, True)) <= tol
    h, us = Q().hessenberg(ad, compute_u=True)
    ho, uso = oracle.hessenberg(a, True)
    assert relerr(n_(h), ho) <= tol
    for x, y in zip(us, uso):
        assert relerr(n_(x), y) <= tol
    hz = np.triu(a, -1)
    q, r = Q().qr_hessenberg(t(hz, dev))
    qo, ro = oracle.qr_hessenberg(hz)
    assert relerr(n_(q), qo) <= tol and relerr(n_(r), ro) <= tol
    assert np.abs(np.einsum('bij,bjk->bik', n_(q).astype(np.float64), n_(r).astype(np.float64)) - hz).max() <= 20 * tol * 4
    rq = n_(Q().rq_hessenberg(t(hz, dev)))
    assert relerr(rq, oracle.rq_hessenberg(hz)) <= tol
    assert relerr(rq, np.einsum('bij,bjk->bik', ro.astype(np.float64), qo.astype(np.float64))) <= 4 * tol
    v = rng.standard_normal((nb, n)).astype(dtype)
    for b in (0, n - 1):
        u, al = Q().householder(t(v, dev), basis=b, return_alpha=True)
        uo, alo = oracle.householder(v, b)
        assert relerr(n_(u), uo) <= tol and relerr(n_(al), alo) <= tol
        # P x = alpha e_b
        px = v.astype(np.float64) - 2 * uo.astype(np.float64) * (uo.astype(np.float64) * v).sum(-1, keepdims=True)
        e = np.zeros((nb, n)); e[:, b] = alo
        if n > 1:
            assert np.abs(px - e).max() <= 20 * tol * np.abs(v).max()


def test_layouts_and_errors(dev, oracle):
    rng = np.random.default_rng(3)
    a = rng.standard_normal((4, 9, 5, 3, 3))
    sym = (a + a.swapaxes(-1, -2)) / 2
    ref = oracle.eig_sym(sym)
    sd = t(sym, dev)
    assert relerr(n_(Q().eig_sym(sd)), ref) <= 1e-11
    # matrix-first ("channel-first") storage: (3, 3, B, X, Y) viewed as (B, X, Y, 3, 3)
    cf = sd.permute(3, 4, 0, 1, 2).contiguous().permute(2, 3, 4, 0, 1)
    assert not cf.is_contiguous()
    assert relerr(n_(Q().eig_sym(cf)), ref) <= 1e-11
    ev, evec = Q().eig_sym(cf[:, ::2], compute_u=True)
    assert relerr(n_(ev), ref[:, ::2]) <= 1e-11 and evec.shape == (4, 5, 5, 3, 3)
    with pytest.raises(ValueError, match='non finite'):
        Q().eig_sym(torch.full((2, 3, 3), float('nan'), device=dev))
    with pytest.raises(ValueError, match='Expected square'):
        Q().eig_sym(torch.zeros(2, 3, 4, device=dev))
    with pytest.raises(TypeError):
        Q().eig_sym(torch.zeros(2, 3, 3, device=dev, dtype=torch.complex64))
    assert Q().eig_sym(torch.zeros(0, 3, 3, device=dev)).shape == (0, 3)
    # zero matrix / diagonal matrix / repeated eigenvalues terminate and are exact
    z = torch.zeros(5, 4, 4, device=dev)
    assert torch.equal(Q().eig_sym(z), torch.zeros(5, 4, device=dev))
    d = torch.diag_embed(torch.tensor([[3.0, 1.0, 2.0, 2.0]], device=dev))
    assert sorted(Q().eig_sym(d)[0].tolist()) == [1.0, 2.0, 2.0, 3.0]
    # fill=False keeps the other triangle of the input
    x = t(rng.standard_normal((6, 4, 4)), dev)
    h = Q().hessenberg_sym(x, upper=True, fill=False)
    assert torch.equal(torch.tril(h, -1), torch.tril(x, -1))
    # givens_apply: per-position coefficients broadcast like upstream, inplace
    c, s = Q().givens(x[:, 0, 0], x[:, 1, 0])
    r = Q().givens_apply(x, c[:, None], s[:, None], 0, 1, side='left')
    assert r[:, 1, 0].abs().max() < 1e-12
    y = x.clone()
    r2 = Q().givens_apply(y, c[:, None], s[:, None], 0, 1, side='left', inplace=True)
    assert r2.data_ptr() == y.data_ptr() and torch.equal(y, r)


def test_large_eig_3x3_properties(dev):
    """Hessian-filter shaped workload: 2e6 symmetric 3x3 fp32, eigenvalues vs the trace /
    determinant invariants and vs torch.linalg.eigvalsh on a sample"""
    n = 2_000_000
    g = torch.Generator(device=dev).manual_seed(11)
    a = torch.randn(n, 3, 3, device=dev, generator=g)
    a = (a + a.transpose(-1, -2)) / 2
    ev = Q().eig_sym(a)
    assert torch.equal(ev, Q().eig_sym(a))
    tr = a.diagonal(0, -1, -2).sum(-1)
    assert ((ev.sum(-1) - tr).abs().max() / tr.abs().max()).item() < 1e-5
    ref = torch.linalg.eigvalsh(a[:4096].double().cpu())
    got = ev[:4096].double().cpu().sort(-1).values
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 5e-6


def test_eig_sym_nothing_left_to_iterate(dev, oracle):
    """zero / diagonal / block-diagonal input and the padding lanes of a ragged last tile have a
    zero off-diagonal from the start: the iteration must stop at once (upstream would spin
    through max_iter = 1024 identical sweeps) and return the diagonal."""
    import nitorch_fastmath_amd as N
    for dtype in (torch.float32, torch.float64):
        z = torch.zeros(5, 4, 4, device=dev, dtype=dtype)
        assert torch.equal(N.eig_sym(z), torch.zeros(5, 4, device=dev, dtype=dtype))
        d = torch.diag_embed(torch.tensor([[3.0, -1.0, 2.0, 0.0]], device=dev, dtype=dtype)).repeat(7, 1, 1)
        v, u = N.eig_sym(d, compute_u=True)
        assert torch.equal(v.sort(-1).values, torch.tensor([-1.0, 0.0, 2.0, 3.0], device=dev, dtype=dtype).expand(7, 4))
        e = oracle.eig_sym(d.cpu().numpy())
        assert np.array_equal(v.cpu().numpy(), e)
        # ragged batch (1000 = 15 * 64 + 40): the 24 idle lanes of the last wavefront must not
        # keep it alive for 1024 sweeps (~1 ms); a converging batch takes tens of microseconds
        g = torch.Generator(device=dev).manual_seed(3)
        a = torch.randn(1000, 3, 3, device=dev, generator=g, dtype=dtype) + 8 * torch.eye(3, device=dev, dtype=dtype)
        s = (a + a.transpose(-1, -2)).contiguous()
        N.eig_sym(s, check_finite=False)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            N.eig_sym(s, check_finite=False)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        assert best < 0.5, f'eig_sym on a ragged batch took {best:.3f} ms'
