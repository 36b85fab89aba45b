"""GPU parity tests for the QR family (givens, householder, hessenberg, qr/rq_hessenberg,
eig_sym) through the C ABI: golden fixtures from the reference, the CPU oracle on seeded
inputs for every order 1..16 (incl. the orders where upstream raises), and properties."""
import os
import numpy as np
import pytest
import torch
from conftest import TOL, EPS, relerr, GOLDEN, parity_ok, within_model

pytestmark = pytest.mark.gpu
QR_NS = (1, 2, 3, 4, 5, 6, 8, 12)


def Q():
    import nitorch_fastmath_amd as N_
    return N_.qr


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def n_(x):
    return x.cpu().numpy()


class GpuQ:
    """the QR family on the GPU behind the oracle's call shapes (numpy in, numpy out), so that
    the golden cases of tests/test_oracle_golden.py run unchanged against the kernels"""

    def __init__(self, dev, arithmetic=None):      # None: the library's default (= 'reference')
        self.dev, self.arithmetic = dev, arithmetic

    def t(self, x):
        return t(x, self.dev)

    def householder(self, v, basis=0):
        u, al = Q().householder(self.t(v), basis=basis, return_alpha=True)
        return n_(u), n_(al)

    def householder_apply(self, a, u, side='both', inverse=False):
        u = [self.t(x) for x in u] if isinstance(u, list) else self.t(u)
        return n_(Q().householder_apply(self.t(a), u, side=side, inverse=inverse))

    def givens_apply(self, a, c, s, i=0, j=None, side='both'):
        if j is None:
            return n_(Q().givens_apply(self.t(a), self.t(c), self.t(s), i, side=side))
        return n_(Q().givens_apply(self.t(a), self.t(c), self.t(s), i, j, side=side))

    def hessenberg(self, a, compute_u=False):
        if compute_u:
            h, us = Q().hessenberg(self.t(a), compute_u=True)
            return n_(h), [n_(x) for x in us]
        return n_(Q().hessenberg(self.t(a)))

    def hessenberg_sym(self, a, upper=True, fill=True, compute_u=False):
        r = Q().hessenberg_sym(self.t(a), upper=upper, compute_u=compute_u)
        return (n_(r[0]), [n_(x) for x in r[1]]) if compute_u else n_(r)

    def qr_hessenberg(self, h):
        q, r = Q().qr_hessenberg(self.t(h))
        return n_(q), n_(r)

    def rq_hessenberg(self, h, u=None):
        if u is None:
            return n_(Q().rq_hessenberg(self.t(h)))
        h2, u2 = Q().rq_hessenberg(self.t(h), self.t(u))
        return n_(h2), n_(u2)

    def eig_sym(self, a, compute_u=False, upper=True):
        r = Q().eig_sym(self.t(a), compute_u=compute_u, upper=upper, arithmetic=self.arithmetic)
        return (n_(r[0]), n_(r[1])) if compute_u else n_(r)


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
def test_golden_family(dev, oracle, golden_qr, dn, n):
    """the reference's own outputs: TOL for float64 and n <= 5, the error model of conftest
    (truth = the float64 oracle on the same inputs) for float32 beyond"""
    from test_oracle_golden import qr_family_cases
    g, k = golden_qr, f'{dn}_n{n}_'
    got = qr_family_cases(GpuQ(dev), g, k, n)
    truth = qr_family_cases(oracle, g, k, n, f64=True) if dn == 'f32' else {}
    for name, val in got.items():
        assert parity_ok(val, g[k + name], n, dn, truth.get(name)), (name, relerr(val, g[k + name]))
    assert not np.isnan(got['hh_u_b0']).any() and len([x for x in got if x.startswith('hess_u')]) == max(n - 2, 0)
    assert parity_ok(n_(Q().hessenberg(t(g[k + 'a'], dev))), g[k + 'hess'], n, dn, truth.get('hess'))
    # eigenvalues of the symmetrised matrix, both arithmetic modes: the golden vector is LAPACK in
    # float64 = the truth of the model
    for mode in ('reference', 'fast'):
        ev = np.sort(GpuQ(dev, mode).eig_sym(g[k + 'sym']), -1)
        assert within_model(ev, g[k + 'eigvalsh'], g[k + 'eigvalsh'], n, dn), (mode, relerr(ev, g[k + 'eigvalsh']))


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', [1, 2, 3, 4, 5])
def test_golden_sym_family(dev, golden_qr, dn, n):
    """orders where the reference itself runs: TOL (1e-6 / 1e-12), no scaling.  The reference-order
    arithmetic reproduces the deflation ORDER and the eigenvector SIGNS of the reference."""
    g, k, tol = golden_qr, f'{dn}_n{n}_', TOL[dn]
    assert Q().SWEEP_ARITHMETIC == 'reference'
    G = GpuQ(dev)                                  # the DEFAULT arithmetic holds the golden order and signs
    a, sym = g[k + 'a'], g[k + 'sym']
    for up in (1, 0):
        tt, us = G.hessenberg_sym(sym, upper=bool(up), compute_u=True)
        assert relerr(tt, g[k + f'hess_sym_{up}']) <= tol
        for i, ui in enumerate(us):
            assert relerr(ui, g[k + f'hess_sym_{up}_u{i}']) <= tol
        assert relerr(G.hessenberg_sym(a, upper=bool(up)), g[k + f'hess_nonsym_{up}']) <= tol
        assert relerr(G.eig_sym(a, upper=bool(up)), g[k + f'eig_{up}']) <= tol
    assert relerr(G.eig_sym(sym), g[k + 'eig']) <= tol        # same deflation ORDER
    ev, evec = G.eig_sym(sym, compute_u=True)
    assert relerr(ev, g[k + 'eig_u_val']) <= tol and relerr(evec, g[k + 'eig_u_vec']) <= 2 * tol
    assert relerr(np.sort(ev, -1), np.sort(g[k + 'eig_batched'], -1)) <= tol
    assert relerr(G.rq_hessenberg(g[k + 'tri']), g[k + 'rq_tri']) <= tol
    eye = np.broadcast_to(np.eye(n, dtype=a.dtype), a.shape).copy()
    h2, u2 = G.rq_hessenberg(g[k + 'tri'], eye)
    assert relerr(u2, g[k + 'rq_tri_u']) <= tol and relerr(h2, g[k + 'rq_tri']) <= tol
    # the fast sweeps: the same eigenvalues as a SET within the error model (order and signs
    # are not specified upstream, `qr.py:45-46`), eigenpairs checked by their defining equations
    F = GpuQ(dev, 'fast')
    truth = np.linalg.eigvalsh(sym.astype(np.float64))
    ev, evec = F.eig_sym(sym, compute_u=True)
    assert within_model(np.sort(ev, -1), np.sort(g[k + 'eig'], -1), truth, n, dn)
    check_eigenpairs(sym, ev, evec, n, dn)


def check_eigenpairs(sym, ev, evec, n, dn):
    """A U = U diag(s) and U^T U = I to c n eps (scaled by |A|)"""
    s64, ev, evec = sym.astype(np.float64), ev.astype(np.float64), evec.astype(np.float64)
    scale = max(np.abs(s64).max(), 1.0)
    res = np.einsum('bij,bjk->bik', s64, evec) - evec * ev[:, None, :]
    assert np.abs(res).max() <= 16 * n * EPS[dn] * scale, np.abs(res).max()
    gram = np.einsum('bji,bjk->bik', evec, evec)
    assert np.abs(gram - np.eye(n)).max() <= 16 * n * EPS[dn]


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16])
def test_vs_oracle(dev, oracle, dn, n):
    """seeded inputs, every order (orders > 5 raise upstream), ragged batch.  The kernels run the
    oracle's operation order, so everything except the fast float32 sweeps is held to TOL against
    the oracle at EVERY order (measured: bit-identical)."""
    dtype = np.float32 if dn == 'f32' else np.float64
    tol = TOL[dn]
    nb = 777 if n <= 8 else 130
    rng = np.random.default_rng(1000 + n)
    a = rng.standard_normal((nb, n, n)).astype(dtype)
    sym = ((a + a.transpose(0, 2, 1)) / 2).astype(dtype)
    ad, sd = t(a, dev), t(sym, dev)
    ref = oracle.eig_sym(sym)
    truth = np.linalg.eigvalsh(sym.astype(np.float64))
    # reference-order arithmetic: the oracle's values in the oracle's order
    ev = n_(Q().eig_sym(sd))                       # default arithmetic, position by position
    assert relerr(ev, ref) <= tol
    assert np.array_equal(ev, n_(Q().eig_sym(sd, arithmetic='reference')))
    # fast sweeps: the same set within the error model (float64: the truth is numpy.linalg in float64,
    # itself good to a few n * eps, which the model's floor covers)
    evf = n_(Q().eig_sym(sd, arithmetic='fast'))
    assert within_model(np.sort(evf, -1), np.sort(ref, -1), truth, n, dn), relerr(np.sort(evf, -1), truth)
    for mode in ('reference', 'fast'):
        ev2, evec = Q().eig_sym(sd, compute_u=True, arithmetic=mode)
        check_eigenpairs(sym, n_(ev2), n_(evec), n, dn)
        assert within_model(np.sort(n_(ev2), -1), np.sort(ref, -1), truth, n, dn)
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
    assert np.abs(np.einsum('bij,bjk->bik', n_(q).astype(np.float64), n_(r).astype(np.float64)) - hz).max() \
        <= 16 * n * EPS[dn] * np.abs(hz).max()
    rq = n_(Q().rq_hessenberg(t(hz, dev)))
    assert relerr(rq, oracle.rq_hessenberg(hz)) <= tol
    assert relerr(rq, np.einsum('bij,bjk->bik', ro.astype(np.float64), qo.astype(np.float64))) <= 4 * n * EPS[dn] + tol
    v = rng.standard_normal((nb, n)).astype(dtype)
    for b in (0, n - 1):
        u, al = Q().householder(t(v, dev), basis=b, return_alpha=True)
        uo, alo = oracle.householder(v, b)
        assert relerr(n_(u), uo) <= tol and relerr(n_(al), alo) <= tol
        # P x = alpha e_b
        px = v.astype(np.float64) - 2 * uo.astype(np.float64) * (uo.astype(np.float64) * v).sum(-1, keepdims=True)
        e = np.zeros((nb, n)); e[:, b] = alo
        if n > 1:
            assert np.abs(px - e).max() <= 16 * n * EPS[dn] * np.abs(v).max()


def test_layouts_and_errors(dev, oracle):
    rng = np.random.default_rng(3)
    a = rng.standard_normal((4, 9, 5, 3, 3))
    sym = (a + a.swapaxes(-1, -2)) / 2
    ref = oracle.eig_sym(sym)
    sd = t(sym, dev)
    assert relerr(n_(Q().eig_sym(sd)), ref) <= 1e-11                                     # default: reference order
    assert relerr(np.sort(n_(Q().eig_sym(sd, arithmetic='fast')), -1), np.sort(ref, -1)) <= 1e-11
    # matrix-first ("channel-first") storage: (3, 3, B, X, Y) viewed as (B, X, Y, 3, 3)
    cf = sd.permute(3, 4, 0, 1, 2).contiguous().permute(2, 3, 4, 0, 1)
    assert not cf.is_contiguous()
    assert relerr(n_(Q().eig_sym(cf, arithmetic='reference')), ref) <= 1e-11
    ev, evec = Q().eig_sym(cf[:, ::2], compute_u=True, arithmetic='reference')
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


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_eig_sym_fast_last_stage_closed_form(dev, dn):
    """the fast sweeps diagonalise the last 2x2 block by one Jacobi rotation (nfm_qr_core.hpp,
    jacobi2_fast1): eigenpairs of 2x2 problems -- random, diagonal (exact), equal diagonal, graded,
    and magnitudes outside the safe range of the hardware rsqrt (those lanes iterate instead)"""
    dtype = torch.float32 if dn == 'f32' else torch.float64
    eps = EPS[dn]
    g = torch.Generator().manual_seed(5)
    a = torch.randn(4099, 2, 2, generator=g, dtype=torch.float64)
    a = a + a.transpose(-1, -2)
    a[0] = torch.tensor([[3.0, 0.0], [0.0, -1.0]])                    # diagonal: comes back bit for bit
    a[1] = torch.tensor([[2.0, 5.0], [5.0, 2.0]])                     # delta = 0: t = +-1
    a[2] = torch.tensor([[1.0, 1e-9], [1e-9, -1.0]])                  # tiny coupling
    a[3] = torch.tensor([[1e-3, 1.0], [1.0, -1e-3]])                  # coupling dominates
    a[4] = torch.zeros(2, 2)
    big, small = (1e18, 1e-18) if dn == 'f32' else (1e140, 1e-140)
    a[5] = torch.tensor([[1.0, 2.0], [2.0, -3.0]], dtype=torch.float64) * big              # delta^2 + b^2 overflows the safe range
    a[6] = torch.tensor([[1.0, 2.0], [2.0, -3.0]], dtype=torch.float64) * small            # ... and underflows it
    ad = a.to(dtype).to(dev)
    v, u = Q().eig_sym(ad, compute_u=True, arithmetic='fast')
    v0 = Q().eig_sym(ad, arithmetic='fast')
    assert torch.equal(v, v0)
    a64 = ad.double().cpu()
    truth = torch.linalg.eigvalsh(a64)
    got = v.double().cpu().sort(-1).values
    scale = a64.abs().amax((-1, -2)).clamp_min(1e-300)
    assert ((got - truth).abs().amax(-1) / scale).max().item() <= 4 * eps
    # eigenpairs: A u = u diag(v), u orthogonal
    ud, vd = u.double().cpu(), v.double().cpu()
    res = (a64 @ ud - ud * vd[:, None, :]).abs().amax((-1, -2)) / scale
    assert res.max().item() <= 8 * eps
    assert (ud.transpose(-1, -2) @ ud - torch.eye(2, dtype=torch.float64)).abs().max().item() <= 8 * eps
    assert sorted(v[0].tolist()) == [-1.0, 3.0] and torch.equal(u[0].abs(), torch.eye(2, dtype=dtype, device=dev))
    assert torch.equal(v[4], torch.zeros(2, dtype=dtype, device=dev))
    # the same stage closes larger problems: bit-identical to itself run twice, and max_iter = 0
    # still means "no iteration at all" (the diagonal of the tridiagonal form, as upstream)
    b = torch.randn(1000, 4, 4, generator=g, dtype=torch.float64)
    b = (b + b.transpose(-1, -2)).to(dtype).to(dev)
    tri = Q().hessenberg_sym(b)
    none = n_(Q().eig_sym(b, max_iter=0, arithmetic='fast'))
    # (a reflector is ill-determined where the column below the diagonal is small, so two roundings of
    # the tridiagonalisation agree only loosely entry by entry)
    assert relerr(none, n_(tri.diagonal(0, -1, -2))) <= 1e4 * eps
    assert relerr(none, n_(Q().eig_sym(b, arithmetic='fast'))) > 1e-2


@pytest.mark.parametrize('n', [3, 4, 6, 8])
def test_eig_sym_fast_tolerance_floor(dev, oracle, n):
    """float32 fast sweeps stop deflating at |e| <= eps/4 |d| (upstream's default 1e-32 asks for
    1e-16 |d|): the eigenvalues stay as close to the float64 truth as the reference-order ones, a
    larger caller tolerance is honoured, and float64 (where 1e-32 IS the working precision) is
    untouched"""
    rng = np.random.default_rng(70 + n)
    a = rng.standard_normal((20000, n, n))
    a = (a + a.swapaxes(-1, -2)).astype(np.float32)
    truth = np.linalg.eigvalsh(a.astype(np.float64))
    scale = np.abs(truth).max(-1, keepdims=True)
    ref = np.sort(oracle.eig_sym(a), -1)
    fast = np.sort(n_(Q().eig_sym(t(a, dev), arithmetic='fast')), -1)
    err_ref = (np.abs(ref - truth) / scale).max()
    err_fast = (np.abs(fast - truth) / scale).max()
    assert err_fast <= 2 * err_ref + 4 * n * EPS['f32'], (err_fast, err_ref)
    # mean error too: the floor must not shift the bulk, not only the worst case
    assert (np.abs(fast - truth) / scale).mean() <= 1.5 * (np.abs(ref - truth) / scale).mean()
    # a loose caller tolerance is kept (errors of its size appear), a tight one is floored
    loose = np.sort(n_(Q().eig_sym(t(a, dev), tol=1e-4, arithmetic='fast')), -1)
    err_loose = (np.abs(loose - truth) / scale).max()
    assert 4 * err_fast < err_loose < 2e-2
    tight = n_(Q().eig_sym(t(a, dev), tol=0.0, arithmetic='fast'))
    assert np.array_equal(np.sort(tight, -1), fast)
    # float64: the default tolerance is above the floor -> same iterates as before the floor existed
    a64 = a[:2000].astype(np.float64)
    f64 = n_(Q().eig_sym(t(a64, dev), arithmetic='fast'))
    assert np.array_equal(f64, n_(Q().eig_sym(t(a64, dev), tol=1e-32, arithmetic='fast')))
    t64 = np.linalg.eigvalsh(a64)
    assert (np.abs(np.sort(f64, -1) - t64) / np.abs(t64).max(-1, keepdims=True)).max() <= 8 * n * EPS['f64']


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', [2, 3, 4, 5, 8])
def test_eig_sym_default_bits_across_ranges(dev, oracle, dn, n):
    """The default arithmetic runs division / square root as the IEEE sequences WITHOUT their range
    handling when a wavefront vote finds every operand in range (nfm_qr_core.hpp, CrRange), and the
    full sequences otherwise: bit-identical to the oracle either way -- matrices scaled across the
    exponent range (in range, out of range at both ends, and both kinds mixed inside one wavefront),
    exact zeros (diagonal / block-diagonal / zero matrices), denormal entries, and non-finite input."""
    dtype = np.float32 if dn == 'f32' else np.float64
    rng = np.random.default_rng(500 + n)
    nb = 64 * 40
    a = rng.standard_normal((nb, n, n))
    a = a + a.swapaxes(-1, -2)
    ex = (-30, -20, -14, -8, 0, 5, 8, 12, 18) if dn == 'f32' else (-250, -190, -120, -40, 0, 20, 45, 70, 140)
    scale = np.ones(nb)
    for i, e in enumerate(ex):                      # whole wavefronts at one scale ...
        scale[64 * i:64 * (i + 1)] = 10.0 ** e
    mixed = slice(64 * len(ex), 64 * (len(ex) + 8))   # ... and wavefronts that mix the scales lane by lane
    scale[mixed] = 10.0 ** rng.choice(ex, size=64 * 8)
    a = a * scale[:, None, None]
    k = 64 * (len(ex) + 8)
    a[k:k + 16] = np.eye(n) * rng.standard_normal((16, 1, n))           # diagonal
    a[k + 16:k + 24] = 0.0                                                # zero
    if n >= 3:                                                            # block diagonal: exact zeros off the blocks
        a[k + 24:k + 40, 0, 1:] = 0.0
        a[k + 24:k + 40, 1:, 0] = 0.0
    tiny = np.finfo(dtype).tiny
    a[k + 40:k + 48] = rng.standard_normal((8, n, n)) * tiny * 4          # denormal-sized entries
    a[k + 40:k + 48] += a[k + 40:k + 48].swapaxes(-1, -2).copy()
    a = a.astype(dtype)
    ref = oracle.eig_sym(a)
    got = n_(Q().eig_sym(t(a, dev)))
    assert np.array_equal(got, ref, equal_nan=True), np.argwhere(~((got == ref) | (np.isnan(got) & np.isnan(ref))))[:8]
    rv, ru = oracle.eig_sym(a, True)
    gv, gu = Q().eig_sym(t(a, dev), compute_u=True)
    assert np.array_equal(n_(gv), rv, equal_nan=True) and np.array_equal(n_(gu), ru, equal_nan=True)
    # non-finite input (check_finite=False): NaN where the oracle has NaN
    b = a[:256].copy()
    b[::7, 0, 0] = np.nan
    b[3::11, -1, 0] = np.inf
    b[3::11, 0, -1] = np.inf
    ref = oracle.eig_sym(b)
    got = n_(Q().eig_sym(t(b, dev), check_finite=False))
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(got[ok], ref[ok])
    # a caller tolerance above one ulp takes the unscreened "stuck" test: same bits
    for tol in (1e-4, 1e-10):
        c = a[:1024]
        assert np.array_equal(n_(Q().eig_sym(t(c, dev), tol=tol)), oracle.eig_sym(c, tol=tol), equal_nan=True)
