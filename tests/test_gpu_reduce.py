"""GPU parity tests for the NaN-omitting reductions."""
import numpy as np
import pytest
import torch
from conftest import TOL

pytestmark = pytest.mark.gpu


def R():
    import nitorch_fastmath_amd as N_
    return N_.reduce


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def same(r, e):
    r, e = float(r), float(e)
    return (np.isnan(r) and np.isnan(e)) or r == e


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('size', [1, 63, 64, 65, 4097, 20011])
@pytest.mark.parametrize('nn', ['nan0', 'nan1', 'nanall'])
def test_golden_full(dev, golden_reduce, dn, size, nn):
    g, k = golden_reduce, f'{dn}_{size}_{nn}_'
    x = g[k + 'x']
    xd = t(x, dev)
    fin = x[np.isfinite(x)]
    scale = max(float(np.abs(fin).sum()), 1e-30)
    for op in ('nansum', 'sum', 'mean'):
        r, e = getattr(R(), op)(xd), g[k + op]
        assert r.dim() == 0 and r.dtype == xd.dtype
        s = scale / (size if op == 'mean' else 1)
        assert same(r, e) or abs(float(r) - float(e)) <= TOL[dn] * s, (op, float(r), float(e))
    r = R().nansum(xd, dtype=torch.float64)
    assert r.dtype == torch.float64
    assert same(r, g[k + 'nansum64']) or abs(float(r) - float(g[k + 'nansum64'])) <= 1e-12 * scale
    for op in ('nanmax', 'nanmin', 'max', 'min'):
        assert same(getattr(R(), op)(xd), g[k + op]), op


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_golden_dims(dev, golden_reduce, dn):
    g = golden_reduce
    x = g[dn + '_nd_x']
    xd = t(x, dev)
    for dim, name in ((0, 'd0'), (1, 'd1'), (2, 'd2'), (-1, 'dm1'), ((0, 2), 'd02'), ((1, 2), 'd12')):
        for fn, key in ((R().nansum, 'nansum'), (R().sum, 'sum'), (R().mean, 'mean')):
            r = fn(xd, dim=dim).cpu().numpy()
            e = g[f'{dn}_nd_{key}_{name}']
            assert r.shape == e.shape
            assert np.array_equal(np.isnan(r), np.isnan(e))
            # a sum is held to |s - s_ref| <= tol * sum|x| of ITS OWN slice (a mean: / count)
            ax = tuple(dim) if isinstance(dim, tuple) else dim
            scale = np.nansum(np.abs(x.astype(np.float64)), axis=ax)
            if key == 'mean':
                scale = scale / np.prod([x.shape[d] for d in (ax if isinstance(ax, tuple) else (ax,))])
            assert (np.abs(np.nan_to_num(r).astype(np.float64) - np.nan_to_num(e)) <= TOL[dn] * scale).all()
        rk = R().nansum(xd, dim=dim, keepdim=True)
        assert tuple(rk.shape) == g[f'{dn}_nd_nansum_keep_{name}'].shape


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', [0, 3, 1000, (1 << 20) + 77, 5_000_011])
def test_vs_oracle_full(dev, oracle, dn, n):
    dtype = np.float32 if dn == 'f32' else np.float64
    rng = np.random.default_rng(n + 1)
    x = rng.standard_normal(n).astype(dtype)
    x[rng.random(n) < 0.01] = np.nan
    for off in (0, 1, 3):      # unaligned base pointers take the scalar head path
        xs = x[off:]
        xd = t(x, dev)[off:]
        if xs.size == 0:
            assert float(R().nansum(xd)) == 0.0
            continue
        scale = max(float(np.nansum(np.abs(xs))), 1e-30)
        r = float(R().nansum(xd, dtype=torch.float64))
        assert abs(r - float(oracle.reduce('nansum', xs, out_f64=True))) <= 1e-12 * scale
        assert abs(float(R().nansum(xd)) - float(oracle.reduce('nansum', xs, out_f64=True))) <= TOL[dn] * scale
        assert float(R().nanmax(xd)) == float(oracle.reduce('nanmax', xs))
        assert float(R().nanmin(xd)) == float(oracle.reduce('nanmin', xs))
        assert np.isnan(float(R().max(xd))) == bool(np.isnan(xs).any())
        # deterministic
        assert float(R().nansum(xd, dtype=torch.float64)) == r


def test_special_values(dev):
    inf = float('inf')
    x = torch.tensor([1.0, float('nan'), -inf, 5.0, inf, float('nan')], device=dev)
    assert float(R().nanmax(x)) == inf and float(R().nanmin(x)) == -inf
    assert np.isnan(float(R().nansum(x)))                 # inf + -inf
    assert float(R().nansum(x[:4])) == -inf
    y = torch.full((1000,), float('nan'), device=dev)
    assert float(R().nansum(y)) == 0.0
    assert float(R().nanmax(y)) == -inf and float(R().nanmin(y)) == inf
    assert np.isnan(float(R().nanmean(y)))
    assert R().nansum(y, keepdim=True).shape == (1,)
    out = torch.empty((), device=dev)
    assert R().nansum(x[:1], out=out).data_ptr() == out.data_ptr() and float(out) == 1.0


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_dim_semantics_vs_numpy(dev, dn):
    """functions that raise upstream (quirks Q10-Q13) follow their documented
    semantics; numpy's nan-functions are the independent check."""
    dtype = np.float32 if dn == 'f32' else np.float64
    rng = np.random.default_rng(11)
    x = rng.standard_normal((5, 70, 6, 33)).astype(dtype)
    x[rng.random(x.shape) < 0.05] = np.nan
    x[2, :, 3, 7] = np.nan                        # an all-NaN reduced slice for dim=1
    xd = t(x, dev)
    tol = 2e-6 if dn == 'f32' else 1e-12
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for dim in (0, 1, 3, -1, (1, 2), (0, 3), (2, 1)):
            ax = dim
            for name, npf in (('nanmax', np.nanmax), ('nanmin', np.nanmin)):
                r = getattr(R(), name)(xd, dim=dim).cpu().numpy()
                e = npf(x, axis=ax)
                fill = -np.inf if name == 'nanmax' else np.inf
                e = np.where(np.isnan(e), fill, e)     # all-NaN -> -inf / +inf like the reference
                assert r.shape == e.shape and np.array_equal(r, e), (name, dim)
            # error model (as for the sums): a mean is held to TOL * mean|x| of its slice,
            # a variance to TOL * mean(x^2), a standard deviation to TOL * sqrt(mean(x^2))
            x64 = x.astype(np.float64)
            m1, m2 = np.nanmean(np.abs(x64), axis=ax), np.nanmean(x64 * x64, axis=ax)
            r = R().nanmean(xd, dim=dim).cpu().numpy()
            e = np.nanmean(x64, axis=ax)
            assert np.array_equal(np.isnan(r), np.isnan(e))
            fin = np.isfinite(e)
            assert (np.abs(r[fin] - e[fin]) <= TOL[dn] * m1[fin]).all()
            for unb in (True, False):
                r = R().nanvar(xd, dim=dim, unbiased=unb).cpu().numpy()
                e = np.nanvar(x64, axis=ax, ddof=int(unb))
                ok = np.isfinite(e)
                assert (np.abs(r[ok] - e[ok]) <= 2 * TOL[dn] * m2[ok]).all()
                r = R().nanstd(xd, dim=dim, unbiased=unb).cpu().numpy()
                assert (np.abs(r[ok] - np.sqrt(e[ok])) <= 2 * TOL[dn] * np.sqrt(m2[ok])).all()
            # NaN-propagating forms
            r = R().max(xd, dim=dim).cpu().numpy()
            e = np.max(x, axis=ax)
            assert np.array_equal(np.isnan(r), np.isnan(e)) and np.array_equal(np.nan_to_num(r), np.nan_to_num(e))
            r = R().var(xd, dim=dim).cpu().numpy()
            e = np.var(x.astype(np.float64), axis=ax, ddof=1)
            assert np.array_equal(np.isnan(r), np.isnan(e))
    # indices: (..., len(dim)) in C order over the reduced dims; scalar dim drops the axis
    v, i = R().nanmax(xd, dim=1, return_indices=True)
    xx = np.where(np.isnan(x), -np.inf, x)
    assert np.array_equal(i.cpu().numpy(), xx.argmax(axis=1))
    v, i = R().nanmin(xd, dim=(1, 2), return_indices=True)
    xx = np.where(np.isnan(x), np.inf, x)
    flat = xx.transpose(0, 3, 1, 2).reshape(5, 33, -1).argmin(-1)
    e = np.stack(np.unravel_index(flat, (70, 6)), -1)
    assert i.shape == (5, 33, 2) and np.array_equal(i.cpu().numpy(), e)
    v, i = R().max(xd, dim=[3], keepdim=True, return_indices=True)
    assert v.shape == (5, 70, 6, 1) and i.shape == (5, 70, 6, 1, 1)
    # first NaN position for the propagating max
    e = np.where(np.isnan(x).any(3), np.isnan(x).argmax(3), np.nan_to_num(x, nan=-np.inf).argmax(3))
    assert np.array_equal(i.cpu().numpy()[..., 0, 0], e)
    # non-contiguous input
    r = R().nansum(xd.transpose(0, 2), dim=1).cpu().numpy()
    e = np.nansum(x.transpose(2, 1, 0, 3).astype(np.float64), axis=1)
    assert (np.abs(r - e) <= TOL[dn] * np.nansum(np.abs(x.transpose(2, 1, 0, 3).astype(np.float64)), axis=1)).all()


def test_large_full_reduction_properties(dev):
    """C4-shaped check at 2^28 elements (1 GiB fp32): split-additivity, permutation
    invariance to rounding, exact max/min, NaN count."""
    n = 1 << 28
    g = torch.Generator(device=dev).manual_seed(4)
    x = torch.randn(n, device=dev, generator=g)
    mask = torch.rand(n, device=dev, generator=g) < 0.01
    x[mask] = float('nan')
    s = float(R().nansum(x, dtype=torch.float64))
    h = n // 2 + 12345
    s2 = float(R().nansum(x[:h], dtype=torch.float64)) + float(R().nansum(x[h:], dtype=torch.float64))
    tot = float(torch.nan_to_num(x).abs().sum(dtype=torch.float64))
    assert abs(s - s2) <= 1e-12 * tot
    ref = float(torch.nan_to_num(x).sum(dtype=torch.float64))
    assert abs(s - ref) <= 1e-12 * tot
    assert abs(float(R().nansum(x)) - ref) <= 1e-6 * tot
    assert float(R().nanmax(x)) == float(torch.nan_to_num(x, nan=-float('inf')).max())
    assert float(R().nanmin(x)) == float(torch.nan_to_num(x, nan=float('inf')).min())
    cnt = float(R().nanmean(x, dtype=torch.float64))
    assert abs(cnt - ref / float((~mask).sum())) <= 1e-12


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_skinny_shapes_split_path(dev, dn):
    """few outputs, long reduced axis: chunked plans (partials folded by the second kernel)"""
    dtype = np.float32 if dn == 'f32' else np.float64
    rng = np.random.default_rng(21)
    tol = 2e-6 if dn == 'f32' else 1e-12
    for shape, dim in (((3, 100_003, 2), 1), ((2, 500_001), 1), ((300_007, 3), 0), ((1, 70_000, 1), 1),
                       ((5, 4096), -1), ((40_000, 2, 2), 0)):
        x = rng.standard_normal(shape).astype(dtype)
        x[rng.random(shape) < 0.02] = np.nan
        xd = t(x, dev)
        x64 = x.astype(np.float64)
        r = R().nansum(xd, dim=dim, dtype=torch.float64).cpu().numpy()
        e = np.nansum(x64, axis=dim)
        scale = np.nansum(np.abs(x64), axis=dim)
        assert r.shape == e.shape and (np.abs(r - e) <= 1e-12 * scale).all(), (shape, dim)
        assert (np.abs(R().nansum(xd, dim=dim).cpu().numpy() - e) <= tol * scale).all()
        assert np.array_equal(R().nanmax(xd, dim=dim).cpu().numpy(), np.nanmax(x, axis=dim))
        assert np.array_equal(R().nanmin(xd, dim=dim).cpu().numpy(), np.nanmin(x, axis=dim))
        m = R().max(xd, dim=dim).cpu().numpy()
        assert np.array_equal(np.isnan(m), np.isnan(x).any(axis=dim))
        v, i = R().nanmax(xd, dim=dim, return_indices=True)
        assert np.array_equal(i.cpu().numpy(), np.where(np.isnan(x), -np.inf, x).argmax(axis=dim))
        mm = R().nanmean(xd, dim=dim, dtype=torch.float64).cpu().numpy()
        assert np.abs(mm - np.nanmean(x64, axis=dim)).max() <= 1e-12 * max(1.0, np.abs(x64[~np.isnan(x64)]).max())


def _check_dim(x, xd, dim, dn):
    """every dim-wise op of one (shape, dim) against numpy's nan-functions"""
    import warnings
    tol = 2e-6 if dn == 'f32' else 1e-12
    x64 = x.astype(np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        e = np.nansum(x64, axis=dim)
        scale = np.nansum(np.abs(x64), axis=dim) + 1e-300
        r = R().nansum(xd, dim=dim, dtype=torch.float64).cpu().numpy()
        assert r.shape == e.shape and (np.abs(r - e) <= 1e-12 * scale).all()
        assert (np.abs(R().nansum(xd, dim=dim).cpu().numpy() - e) <= tol * scale).all()
        s = R().sum(xd, dim=dim).cpu().numpy()
        assert np.array_equal(np.isnan(s), np.isnan(x).any(axis=dim))
        for name, fill, arg in (('nanmax', -np.inf, 'argmax'), ('nanmin', np.inf, 'argmin')):
            xx = np.where(np.isnan(x), fill, x)
            ev = getattr(xx, name[3:])(axis=dim)
            assert np.array_equal(getattr(R(), name)(xd, dim=dim).cpu().numpy(), ev), name
            v, i = getattr(R(), name)(xd, dim=dim, return_indices=True)
            assert np.array_equal(v.cpu().numpy(), ev), name
            assert np.array_equal(i.cpu().numpy(), getattr(xx, arg)(axis=dim)), name + ' index'
        v, i = R().max(xd, dim=dim, return_indices=True)
        anyn = np.isnan(x).any(axis=dim)
        assert np.array_equal(np.isnan(v.cpu().numpy()), anyn)
        ei = np.where(anyn, np.isnan(x).argmax(axis=dim), np.nan_to_num(x, nan=-np.inf).argmax(axis=dim))
        assert np.array_equal(i.cpu().numpy(), ei), 'first-NaN index'
        big = max(1.0, float(np.nanmax(np.abs(x64)))) if np.isfinite(x64).any() else 1.0
        m, em = R().nanmean(xd, dim=dim, dtype=torch.float64).cpu().numpy(), np.nanmean(x64, axis=dim)
        assert np.array_equal(np.isnan(m), np.isnan(em)) and np.nanmax(np.abs(m - em), initial=0) <= 1e-12 * big
        m = R().mean(xd, dim=dim).cpu().numpy()
        assert np.array_equal(np.isnan(m), anyn)
        for unb in (True, False):
            r = R().nanvar(xd, dim=dim, unbiased=unb, dtype=torch.float64).cpu().numpy()
            ev = np.nanvar(x64, axis=dim, ddof=int(unb))
            ok = np.isfinite(ev)
            assert np.array_equal(np.isnan(r[~ok]), np.isnan(ev[~ok]))
            assert np.abs(r[ok] - ev[ok]).max(initial=0) <= 1e-11 * big * big
            r = R().nanstd(xd, dim=dim, unbiased=unb).cpu().numpy()
            m2 = np.nanmean(x64 * x64, axis=dim)
            assert (np.abs(r[ok] - np.sqrt(ev[ok])) <= 2 * TOL[dn] * np.sqrt(m2[ok])).all()      # TOL * sqrt(mean x^2)


# (shape, dim): every plan of nfm_reduce_dim.hip -- GROUP slabs (vector and scalar, every group
# size), FLAT slabs (inner == 1, power-of-two / odd / large small-inner, ragged ends, chunked),
# COL (vector and scalar, chunked and not)
_PLAN_SHAPES = [
    ((1000, 1), 1), ((1000, 2), 1), ((999, 3), 1), ((1000, 4), 1), ((777, 8), 1), ((513, 12), 1),
    ((300, 32), 1), ((129, 100), 1), ((67, 256), 1), ((67, 255), 1), ((50, 64), 1), ((50, 65), 1),
    ((33, 1000), 1), ((9, 4096), 1), ((3, 100_001), 1), ((1, 1_000_003), 1), ((20_000, 260), 1),
    ((5, 700, 2), 1), ((5, 701, 3), 1), ((4, 300, 4), 1), ((3, 1000, 5), 1), ((7, 130, 8), 1),
    ((2, 5000, 10), 1), ((1, 100_000, 16), 1), ((3, 333, 37), 1), ((2, 250, 64), 1), ((2, 90, 100), 1),
    ((3, 70, 128), 1), ((2, 40, 256), 1), ((1, 200_001, 3), 1), ((600, 4, 4), 1), ((10_000, 3, 3), 1),
    ((63, 3, 128), 1), ((2, 300, 1000), 1), ((3, 50, 1001), 1), ((1, 7, 100_000), 1), ((1, 3000, 260), 1),
    ((1, 5000, 515), 1), ((2, 8, 65_536), 1), ((1, 100_000, 300), 1),
]


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_dim_kernel_plans(dev, dn):
    dtype = np.float32 if dn == 'f32' else np.float64
    rng = np.random.default_rng(77)
    for shape, dim in _PLAN_SHAPES:
        x = rng.standard_normal(shape).astype(dtype)
        x[rng.random(shape) < 0.03] = np.nan
        x[rng.random(shape) < 0.05] = 1.5            # ties: first occurrence must win
        if x.shape[0] > 1:
            x[1].fill(np.nan)                         # all-NaN slices
        _check_dim(x, t(x, dev), dim, dn)
        if x.size < 300_000:                          # misaligned base pointer: scalar plans
            flat = np.concatenate([np.zeros(1, dtype), x.reshape(-1)])
            xd = t(flat, dev)[1:].reshape(shape)
            _check_dim(x, xd, dim, dn)


def test_dim_reductions_are_reproducible(dev):
    g = torch.Generator(device=dev).manual_seed(5)
    for shape in ((16, 1 << 20), (1, 1 << 18, 12), (1 << 12, 300), (2, 1 << 16, 512)):
        x = torch.randn(shape, device=dev, generator=g)
        a = R().nansum(x, dim=1)
        for _ in range(3):
            assert torch.equal(a, R().nansum(x, dim=1))
        v = R().nanvar(x, dim=1)
        assert torch.equal(v, R().nanvar(x, dim=1))


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_non_adjacent_dims_staged(dev, dn):
    """batch + spatial dims of a channel-first field: reduced run by run, no permuting copy"""
    import warnings
    dtype = np.float32 if dn == 'f32' else np.float64
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((3, 4, 9, 11, 5)) + 50.0).astype(dtype)
    x[rng.random(x.shape) < 0.05] = np.nan
    x[:, 2] = np.nan                               # one all-NaN channel
    xd = t(x, dev)
    x64 = x.astype(np.float64)
    tol = 2e-6 if dn == 'f32' else 1e-12
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for dim in ((0, 2, 3, 4), (0, 2), (1, 3), (0, 4), (4, 0, 2)):
            ax = tuple(dim)
            e = np.nansum(x64, axis=ax)
            scale = np.nansum(np.abs(x64), axis=ax) + 1e-300
            r = R().nansum(xd, dim=dim).cpu().numpy()
            assert r.shape == e.shape and (np.abs(r - e) <= tol * scale).all(), dim
            assert R().nansum(xd, dim=dim, keepdim=True).shape == np.nansum(x64, axis=ax, keepdims=True).shape
            assert np.array_equal(np.isnan(R().sum(xd, dim=dim).cpu().numpy()), np.isnan(x).any(axis=ax))
            assert np.array_equal(R().nanmax(xd, dim=dim).cpu().numpy(),
                                  np.where(np.isnan(x), -np.inf, x).max(axis=ax))
            assert np.array_equal(R().nanmin(xd, dim=dim).cpu().numpy(),
                                  np.where(np.isnan(x), np.inf, x).min(axis=ax))
            m = R().nanmean(xd, dim=dim, dtype=torch.float64).cpu().numpy()
            em = np.nanmean(x64, axis=ax)
            assert np.array_equal(np.isnan(m), np.isnan(em)) and np.nanmax(np.abs(m - em), initial=0) <= 1e-10
            assert np.array_equal(np.isnan(R().mean(xd, dim=dim).cpu().numpy()), np.isnan(x).any(axis=ax))
            for unb in (True, False):
                v = R().nanvar(xd, dim=dim, unbiased=unb, dtype=torch.float64).cpu().numpy()
                ev = np.nanvar(x64, axis=ax, ddof=int(unb))
                ok = np.isfinite(ev)
                assert np.array_equal(np.isnan(v[~ok]), np.isnan(ev[~ok]))
                assert np.abs(v[ok] - ev[ok]).max(initial=0) <= 1e-9
                sd = R().nanstd(xd, dim=dim, unbiased=unb, keepdim=True)
                assert sd.shape == np.nanstd(x64, axis=ax, keepdims=True).shape
                m2 = np.nanmean(x64 * x64, axis=ax)
                assert (np.abs(sd.cpu().numpy().reshape(ev.shape)[ok] - np.sqrt(ev[ok])) <= 2 * TOL[dn] * np.sqrt(m2[ok])).all()


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_permuted_contiguous_inputs_reduce_in_place(dev, dn):
    """channel-last views of channel-first fields (a dim permutation of a contiguous tensor)
    are reduced through the mapped dims of the underlying tensor: no copy, same results"""
    import warnings
    dtype = np.float32 if dn == 'f32' else np.float64
    rng = np.random.default_rng(9)
    base = rng.standard_normal((2, 6, 7, 9, 5)).astype(dtype)          # (B, C, X, Y, Z)
    base[rng.random(base.shape) < 0.05] = np.nan
    bd = t(base, dev)
    for perm in ((0, 2, 3, 4, 1), (1, 0, 2, 3, 4), (4, 3, 2, 1, 0), (0, 1, 3, 2, 4)):
        x, xd = base.transpose(perm), bd.permute(perm)
        assert not xd.is_contiguous()
        x64 = x.astype(np.float64)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for dim in (-1, 0, 2, (1, 2), (1, 2, 3), (0, 4), [3]):
                ax = tuple(dim) if isinstance(dim, (list, tuple)) else dim
                e = np.nansum(x64, axis=ax)
                r = R().nansum(xd, dim=dim, dtype=torch.float64).cpu().numpy()
                assert r.shape == e.shape and np.abs(r - e).max() <= 1e-12 * np.nansum(np.abs(x64)), (perm, dim)
                rk = R().nansum(xd, dim=dim, keepdim=True)
                assert rk.shape == np.nansum(x64, axis=ax, keepdims=True).shape
                xx = np.where(np.isnan(x), -np.inf, x)
                assert np.array_equal(R().nanmax(xd, dim=dim).cpu().numpy(), xx.max(axis=ax)), (perm, dim)
                m = R().nanmean(xd, dim=dim, dtype=torch.float64).cpu().numpy()
                assert np.nanmax(np.abs(m - np.nanmean(x64, axis=ax)), initial=0) <= 1e-12
                v = R().nanvar(xd, dim=dim, dtype=torch.float64).cpu().numpy()
                ev = np.nanvar(x64, axis=ax, ddof=1)
                ok = np.isfinite(ev)
                assert np.abs(v[ok] - ev[ok]).max(initial=0) <= 1e-11
            # indices: scalar dim and a pair of dims
            v, i = R().nanmax(xd, dim=-1, return_indices=True)
            assert np.array_equal(i.cpu().numpy(), np.where(np.isnan(x), -np.inf, x).argmax(-1))
            v, i = R().nanmin(xd, dim=(1, 2), return_indices=True)
            xx = np.where(np.isnan(x), np.inf, x)
            sh = x.shape
            flat = np.moveaxis(xx, (1, 2), (-2, -1)).reshape(sh[0], sh[3], sh[4], -1).argmin(-1)
            assert np.array_equal(i.cpu().numpy(), np.stack(np.unravel_index(flat, (sh[1], sh[2])), -1)), perm


def test_inplace_never_modifies_the_input(dev):
    """`inplace=True` is a permission upstream ("Allow modifying the input tensor in-place",
    `reduce.py:72-74`).  The reference uses it: `nansum(x, inplace=True)` leaves zeros, `nanmax` /
    `nanmin` leave -inf / +inf where the caller's tensor held NaN (`reduce.py:502-509`, `:258-260`;
    probed).  This backend masks inside the kernel and never writes to its input: after
    every call the caller's tensor is BIT-identical, NaNs included.  A caller that relied on the
    side effect must do `x.nan_to_num_(0)` itself."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(257, 33, generator=g)
    x[torch.rand(257, 33, generator=g) < 0.1] = float('nan')
    xd = x.to(dev)
    bits = xd.view(torch.int32).clone()
    for fn in (R().nansum, R().nanmax, R().nanmin, R().nanmean, R().nanvar, R().nanstd):
        for dim in (None, 0, 1, (0, 1)):
            fn(xd, dim=dim, inplace=True)
            assert torch.equal(xd.view(torch.int32), bits), (fn.__name__, dim)
    for fn in (R().sum, R().max, R().min, R().mean):
        fn(xd, dim=1, omitnan=True, inplace=True)
        assert torch.equal(xd.view(torch.int32), bits), fn.__name__
    # and the values are those of the non-inplace call
    assert torch.equal(R().nansum(xd, dim=1, inplace=True), R().nansum(xd, dim=1))
    assert torch.equal(R().nanmax(xd, dim=0, inplace=True), R().nanmax(xd, dim=0))


def test_median_semantics(dev):
    """quirk Q14: the reference's `median` docstring says "always omits NaNs", its code calls
    `torch.median`, which propagates them.  Here `omitnan=False` (default) = the reference's actual
    result, `omitnan=True` = the documented intent; index layout as for max / min; the index is the
    FIRST position holding the median value (torch picks the position a stable sort would put at the
    median's rank: the same value, possibly another of its duplicates)."""
    x = torch.tensor([[3., float('nan'), 1., 2., 5.], [4., 1., 3., 2., 0.], [float('nan')] * 5], device=dev)
    m = R().median(x, dim=1)
    assert torch.isnan(m[0]) and m[1] == 2 and torch.isnan(m[2])                   # = torch.median
    assert torch.equal(torch.isnan(m), torch.isnan(torch.median(x.cpu(), dim=1).values.to(dev)))
    mo, io = R().median(x, dim=1, omitnan=True, return_indices=True)
    assert mo[0] == 2 and mo[1] == 2 and torch.isnan(mo[2])                        # lower median of {1,2,3,5}
    assert io[0] == 3 and io[1] == 3 and io[2] == 0 and io.dtype == torch.long
    mn, i_n = R().median(x, dim=1, return_indices=True)
    assert i_n[0] == 1 and i_n[2] == 0                                             # the first NaN
    assert R().median(x[1]) == 2 and torch.isnan(R().median(x))
    assert R().median(x, omitnan=True) == 2                                        # {0,1,1,2,2,3,3,4,5} -> 2
    y = torch.arange(24., device=dev).reshape(2, 3, 4)
    v, i = R().median(y, dim=(0, 2), return_indices=True)
    assert v.tolist() == [3., 7., 11.] and i.shape == (3, 2) and i[0].tolist() == [0, 3]
    vk = R().median(y, dim=(0, 2), keepdim=True)
    assert vk.shape == (1, 3, 1)
    with pytest.raises(IndexError):
        R().median(torch.zeros(3, 0, device=dev), dim=1)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('red', [1, 2, 3, 27, 63, 64, 65, 127, 128, 129, 500, 1023, 1024, 1025, 4097, 70001])
def test_median_vs_oracle_and_torch(dev, oracle, dn, red):
    """every row-length regime of the radix selection (1 / 2 / 4 / 8 / 16 keys per lane, the
    multi-pass histogram form beyond 1024), NaNs, signed zeros, infinities, duplicates; values equal
    to numpy's sort and to torch.median / nanmedian on the CPU, bit for bit"""
    dtype = np.float32 if dn == 'f32' else np.float64
    rng = np.random.default_rng(red)
    rows = 37 if red <= 1025 else 5
    x = rng.standard_normal((rows, red)).astype(dtype)
    x[0] = np.round(x[0] * 2) / 2                      # many duplicates
    if red > 2:
        x[1, ::3] = np.nan
        x[2, 0], x[2, -1], x[2, red // 2] = np.inf, -np.inf, -0.0
        x[3] = np.nan                                   # all NaN
        x[4, 1] = np.nan
    xd = t(x, dev)
    for omit in (False, True):
        v, i = R().median(xd, dim=1, omitnan=omit, return_indices=True)
        rv, ri = oracle.median(x, 1, omitnan=omit)
        v, i = v.cpu().numpy(), i.cpu().numpy()
        assert np.array_equal(np.isnan(v), np.isnan(rv)) and np.array_equal(v[~np.isnan(v)], rv[~np.isnan(rv)]), omit
        assert np.array_equal(i, ri), omit
        tv = (torch.nanmedian if omit else torch.median)(torch.from_numpy(x), dim=1).values.numpy()
        assert np.array_equal(np.isnan(v), np.isnan(tv)) and np.array_equal(v[~np.isnan(v)], tv[~np.isnan(tv)])
        ok = ~np.isnan(v)
        assert np.array_equal(x[np.arange(rows)[ok], i[ok]].view(np.uint8), v[ok].view(np.uint8))   # the index holds the value, bit for bit
    # columns (reduced dim first) and the whole tensor
    v0 = R().median(xd.t().contiguous(), dim=0, omitnan=True).cpu().numpy()
    assert np.array_equal(np.nan_to_num(v0, nan=7.0), np.nan_to_num(oracle.median(x, 1, omitnan=True)[0], nan=7.0))
    flat = R().median(xd, omitnan=True)
    assert float(flat) == float(oracle.median(x.reshape(-1), None, omitnan=True)[0])


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_median_row_per_lane(dev, oracle, dn):
    """rows of 2..128 (float64: 2..64) elements in batches of >= 4096 rows: one row per lane, sorted in registers by
    the merge-exchange network (median_lane_kernel) -- every length, ragged last tile, NaNs (kept and
    omitted), signed zeros, infinities, ties; values and first-position indices equal to the oracle's,
    bit for bit, and to the few-rows kernel's on the same rows"""
    dtype = np.float32 if dn == 'f32' else np.float64
    for red in range(2, 130 if dn == 'f32' else 66):      # one past the limit: the wavefront-per-row kernel
        rng = np.random.default_rng(1000 + red)
        rows = 4096 + 3 * 256 + 41
        x = rng.standard_normal((rows, red)).astype(dtype)
        x[::7] = np.round(x[::7] * 2) / 2                                   # ties
        x[rng.random((rows, red)) < 0.03] = np.nan
        x[5] = np.nan                                                       # all NaN
        x[6, 0], x[6, -1] = np.inf, -np.inf
        x[8] = 0.0
        x[8, ::2] = -0.0                                                    # signed zeros only
        xd = t(x, dev)
        for omit in (False, True):
            v, i = R().median(xd, dim=1, omitnan=omit, return_indices=True)
            v0 = R().median(xd, dim=1, omitnan=omit)
            rv, ri = oracle.median(x, 1, omitnan=omit)
            v, i, v0 = v.cpu().numpy(), i.cpu().numpy(), v0.cpu().numpy()
            assert np.array_equal(v.view(np.uint8), v0.view(np.uint8))      # with and without the index output
            nn = ~np.isnan(rv)
            assert np.array_equal(np.isnan(v), np.isnan(rv)), (red, omit)
            assert np.array_equal(v[nn].view(np.uint8), rv[nn].view(np.uint8)), (red, omit)
            assert np.array_equal(i, ri), (red, omit)
            few = R().median(xd[:100], dim=1, omitnan=omit, return_indices=True)     # < 4096 rows: the group kernel
            assert np.array_equal(few[0].cpu().numpy().view(np.uint8), v[:100].view(np.uint8))
            assert np.array_equal(few[1].cpu().numpy(), i[:100])


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_median_padded_row_per_lane(dev, oracle, dn):
    """rows of 129..192 (float64: 65..96) elements in batches of >= 32768 rows: one PADDED row per lane
    (median_lane_pad_kernel: the row is padded to the next bucket length with keys that sort last); rows of 193..256
    (97..128): TWO lanes per row (median_lane_pair_kernel: each sorts a half, the halves meet in a bitonic split) -- both
    ends and the middle of every bucket, NaNs kept and omitted, all-NaN rows, infinities, signed zeros, ties;
    values and first-position indices equal to the oracle's bit for bit, and to the few-rows kernel's"""
    dtype = np.float32 if dn == 'f32' else np.float64
    lo, step, nb = (128, 16, 4) if dn == 'f32' else (64, 8, 4)
    lengths = sorted({lo + b * step + d for b in range(nb) for d in (1, step // 2, step)}
                     | {lo + nb * step + 1, lo + nb * step + 7, 2 * lo - 9, 2 * lo - 1, 2 * lo}    # two lanes per row
                     | {2 * lo + 1})
    for red in lengths:                                   # the last one is past the limit: the wavefront-per-row kernel
        rng = np.random.default_rng(3000 + red)
        rows = 32768 + 64 + 29
        x = rng.standard_normal((rows, red)).astype(dtype)
        x[::7] = np.round(x[::7] * 2) / 2                                   # ties
        x[rng.random((rows, red)) < 0.02] = np.nan
        x[5] = np.nan                                                       # all NaN
        x[6, 0], x[6, -1] = np.inf, -np.inf
        x[8] = 0.0
        x[8, ::2] = -0.0                                                    # signed zeros only
        x[9, :-1] = np.nan                                                  # one value left
        xd = t(x, dev)
        for omit in (False, True):
            v, i = R().median(xd, dim=1, omitnan=omit, return_indices=True)
            v0 = R().median(xd, dim=1, omitnan=omit)
            rv, ri = oracle.median(x, 1, omitnan=omit)
            v, i, v0 = v.cpu().numpy(), i.cpu().numpy(), v0.cpu().numpy()
            assert np.array_equal(v.view(np.uint8), v0.view(np.uint8))      # with and without the index output
            nn = ~np.isnan(rv)
            assert np.array_equal(np.isnan(v), np.isnan(rv)), (red, omit)
            assert np.array_equal(v[nn].view(np.uint8), rv[nn].view(np.uint8)), (red, omit)
            assert np.array_equal(i, ri), (red, omit)
            few = R().median(xd[:100], dim=1, omitnan=omit, return_indices=True)     # few rows: the group kernel
            assert np.array_equal(few[0].cpu().numpy().view(np.uint8), v[:100].view(np.uint8))
            assert np.array_equal(few[1].cpu().numpy(), i[:100])


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_median_middle_dim_without_transpose(dev, oracle, dn):
    """the channel dim (or a block of adjacent dims) of a contiguous channel-first field: one row per
    lane straight from the (outer, red, inner) layout (nfm_reduce_median_mid) -- same values and indices
    as moving the dims last first, which is what the other path (and upstream) does"""
    dtype = np.float32 if dn == 'f32' else np.float64
    rng = np.random.default_rng(77)
    for shape, dim in (((3, 27, 40, 41), 1), ((2, 5, 5, 61, 37), (1, 2)), ((70, 9, 67), 1), ((4100, 2, 3), 1),
                       ((2, 64 if dn == 'f64' else 128, 4099), 1), ((5, 3, 4, 1000), (1, 2))):
        x = rng.standard_normal(shape).astype(dtype)
        x[rng.random(shape) < 0.02] = np.nan
        xd = t(x, dev)
        dims = (dim,) if isinstance(dim, int) else dim
        last = np.moveaxis(x, dims, tuple(range(-len(dims), 0)))
        flat = np.ascontiguousarray(last).reshape(-1, int(np.prod([shape[d] for d in dims])))
        for omit in (False, True):
            v, i = R().median(xd, dim=dim, omitnan=omit, return_indices=True)
            rv, ri = oracle.median(flat, 1, omitnan=omit)
            v, i = v.cpu().numpy(), i.cpu().numpy()
            rv = rv.reshape(v.shape)
            nn = ~np.isnan(rv)
            assert np.array_equal(np.isnan(v), np.isnan(rv)), (shape, omit)
            assert np.array_equal(v[nn].view(np.uint8), rv[nn].view(np.uint8)), (shape, omit)
            sub = np.stack(np.unravel_index(ri, [shape[d] for d in dims]), -1).reshape(v.shape + (len(dims),))
            assert np.array_equal(i.reshape(sub.shape if len(dims) > 1 else v.shape),
                                  sub if len(dims) > 1 else sub[..., 0]), (shape, omit)
            # the same call on a transposed copy takes the rows path: identical
            v2 = R().median(t(np.ascontiguousarray(last), dev), dim=tuple(range(-len(dims), 0)), omitnan=omit)
            assert np.array_equal(v2.cpu().numpy().view(np.uint8), v.view(np.uint8))
    # the channel-last VIEW of a channel-first field: reduced in place through the contiguous tensor
    x = rng.standard_normal((2, 9, 70, 71)).astype(dtype)
    xd = t(x, dev)
    view = xd.movedim(1, -1)
    assert not view.is_contiguous()
    for keep in (False, True):
        v, i = R().median(view, dim=-1, keepdim=keep, return_indices=True)
        rv, ri = R().median(view.contiguous(), dim=-1, keepdim=keep, return_indices=True)
        assert v.shape == rv.shape and torch.equal(v, rv) and torch.equal(i, ri)
    assert torch.equal(R().median(view, dim=0), R().median(view.contiguous(), dim=0))
    # backward: the gradient goes to the selected element
    xg = t(rng.standard_normal((2, 9, 5000)).astype(dtype), dev).requires_grad_(True)
    R().median(xg, dim=1).sum().backward()
    assert float(xg.grad.sum()) == 2 * 5000 and int((xg.grad != 0).sum()) == 2 * 5000


def test_median_large(dev):
    """2^27 elements in one row (the dim=None form of a volume): 4 streaming passes; against a full
    device sort (torch.nanmedian on the device returned another element for this input), plus a 3-D
    field reduced over its spatial dims"""
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(1 << 27, device=dev, generator=g)
    x[::1001] = float('nan')
    got = R().median(x, omitnan=True)
    valid = x[~torch.isnan(x)]
    truth = torch.sort(valid).values[(valid.numel() - 1) // 2]          # the lower median by a full sort
    assert got == truth, (float(got), float(truth), float(torch.nanmedian(x)))
    assert torch.isnan(R().median(x))
    y = x[: (1 << 26) + 12345].clone().nan_to_num_(0.25)                   # odd length, no NaN, one value repeated
    assert R().median(y) == torch.sort(y).values[(y.numel() - 1) // 2]
    f = torch.randn(3, 8, 200, 200, device=dev, generator=g)
    v, i = R().median(f, dim=(2, 3), return_indices=True)
    ref = torch.median(f.reshape(3, 8, -1), dim=-1).values
    assert torch.equal(v, ref) and i.shape == (3, 8, 2)
    assert torch.equal(f[torch.arange(3)[:, None], torch.arange(8)[None], i[..., 0], i[..., 1]], v)
