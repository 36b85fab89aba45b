"""BASELINE.json's configurations at FULL size, checked through size-independent properties
(round trips, linearity, split-additivity, bitwise repeatability) plus sampled oracle parity.
C2: 1e8 4x4 fp32 solves; C3: 1e7 8x8 fp64 inverses; C4: 2^33 fp32 elements (32 GiB) nansum /
nanmax; C5: the per-GPU share, 1e8 6x6 fp32 solves."""
import numpy as np
import pytest
import torch
from conftest import relerr

pytestmark = pytest.mark.gpu


def _need(dev, gib):
    free, _ = torch.cuda.mem_get_info(dev)
    if free < gib * (1 << 30):
        pytest.skip(f'needs {gib} GiB of free HBM')


@pytest.mark.parametrize('M,cfg', [(4, 'C2'), (6, 'C5 (one GPU share)')])
def test_sym_solve_full_size(dev, oracle, M, cfg):
    import nitorch_fastmath_amd as N
    from bench import spd_compact
    n = 100_000_000
    _need(dev, 24 if M == 6 else 16)
    mat, v = spd_compact(n, M, torch.float32, dev, 7)
    y = N.sym_matvec(mat, v)                       # b = A v
    x = N.sym_solve(mat, y)                        # x = A^-1 b
    d = (x - v).abs_()
    assert (d.amax() / v.abs().amax()).item() < 5e-5          # cond <~ 10, fp32
    del d
    assert torch.equal(x, N.sym_solve(mat, y))                # bitwise repeatable
    y.mul_(2)
    x2 = N.sym_solve(mat, y)
    assert torch.equal(x2, x.mul_(2))                         # exact under power-of-two scaling
    del x2
    # sampled oracle parity: first tile, an interior stretch, the ragged tail
    idx = torch.cat([torch.arange(0, 2048), torch.arange(n // 3, n // 3 + 2048), torch.arange(n - 999, n)]).to(dev)
    ref = oracle.sym_solve(mat[idx].cpu().numpy(), y[idx].cpu().numpy())
    got = x[idx].cpu().numpy()
    if M <= 4:
        assert np.array_equal(got, ref)                       # closed forms: bit-exact
    else:
        assert relerr(got, ref) <= 1e-6


def test_batchinv_full_size_c3(dev, oracle):
    import nitorch_fastmath_amd as N
    n = 10_000_000
    _need(dev, 24)
    g = torch.Generator(device=dev).manual_seed(11)
    a = torch.randn(n, 8, 8, device=dev, generator=g, dtype=torch.float64)
    a.diagonal(dim1=-2, dim2=-1).add_(8.0)                    # N(0,1) + 8 I
    inv = N.batchinv(a)
    assert torch.equal(inv, N.batchinv(a))                    # bitwise repeatable
    v = torch.randn(n, 8, device=dev, generator=g, dtype=torch.float64)
    w = N.batchmatvec(inv, N.batchmatvec(a, v))               # A^-1 (A v) == v
    assert ((w - v).abs_().amax() / v.abs().amax()).item() < 1e-11
    del w, v
    back = N.batchinv(inv)                                    # inv(inv(A)) == A
    assert ((back - a).abs_().amax() / a.abs().amax()).item() < 1e-11
    del back
    idx = torch.cat([torch.arange(0, 1024), torch.arange(n - 333, n)]).to(dev)
    assert relerr(inv[idx].cpu().numpy(), oracle.batch_inv(a[idx].cpu().numpy())) <= 1e-12


def test_reductions_full_size_c4(dev):
    import nitorch_fastmath_amd as N
    R = N.reduce
    n = 1 << 33                                               # 32 GiB of fp32
    _need(dev, 80)
    x = torch.empty(n, device=dev)
    g = torch.Generator(device=dev).manual_seed(4)
    chunk = 1 << 28
    for lo in range(0, n, chunk):
        part = x[lo:lo + chunk]
        part.normal_(generator=g)
        part[torch.rand(chunk, device=dev, generator=g) < 0.01] = float('nan')
    # checksum of checksums: the full sum equals the sum of its pieces (fp64 accumulation)
    s = float(R.nansum(x, dtype=torch.float64))
    assert s == float(R.nansum(x, dtype=torch.float64))       # fixed geometry: reproducible
    parts, absum, mx, mn, cnt = 0.0, 0.0, -np.inf, np.inf, 0
    for lo in range(0, n, chunk):
        p = x[lo:lo + chunk]
        q = torch.nan_to_num(p)
        parts += float(q.sum(dtype=torch.float64))
        absum += float(q.abs_().sum(dtype=torch.float64))
        mx = max(mx, float(torch.nan_to_num(p, nan=-float('inf')).max()))
        mn = min(mn, float(torch.nan_to_num(p, nan=float('inf')).min()))
        cnt += int((p == p).sum())
        del q
    assert abs(s - parts) <= 1e-12 * absum
    assert abs(float(R.nansum(x)) - parts) <= 1e-6 * absum
    assert float(R.nanmax(x)) == mx and float(R.nanmin(x)) == mn
    assert abs(float(R.nanmean(x, dtype=torch.float64)) - parts / cnt) <= 1e-12
    assert np.isnan(float(R.sum(x))) and np.isnan(float(R.max(x)))
    # the same bytes viewed as a matrix: dim-wise results must fold back to the full ones
    xm = x.view(1 << 13, 1 << 20)
    rows = R.nansum(xm, dim=1, dtype=torch.float64)
    assert abs(float(rows.sum()) - s) <= 1e-12 * absum
    cols = R.nansum(xm, dim=0, dtype=torch.float64)
    assert abs(float(cols.sum()) - s) <= 1e-12 * absum
    v, i = R.nanmax(xm, dim=1, return_indices=True)
    assert float(v.max()) == mx
    assert torch.equal(xm.gather(1, i.unsqueeze(1)).squeeze(1), v)    # the index points at the value


def test_median_of_a_row_longer_than_2_pow_32(dev):
    """`median(x)` with dim=None on more than 2^32 elements (the C4 tensor has 2^33): the radix selection of a
    long row counts in 64 bits.  Constant data is the hard case -- every element of a pass lands in ONE
    histogram bin, whose count (and the scan over the bins) passes 2^32; 32-bit counters wrapped and
    returned another element."""
    from nitorch_fastmath_amd import reduce as R
    n = (1 << 32) + 12345
    _need(dev, 24)
    x = torch.zeros(n, device=dev)
    x[:1000] = -1.0
    x[-7:] = 2.0
    assert float(R.median(x)) == 0.0
    v, i = R.median(x, dim=0, return_indices=True)            # (dim=None returns the value only, like upstream)
    assert float(v) == 0.0 and int(i) == 1000                 # first position holding the value
    x[1000:1000 + (1 << 31)] = -3.0                           # now 2^31 + 1000 of 2^32 + 12345 are below 0: still 0
    assert float(R.median(x)) == 0.0
    x[1000:1000 + (1 << 31) + 8000] = -3.0                    # more than half below: the median moves
    assert float(R.median(x)) == -3.0
    x[5] = float('nan')
    assert torch.isnan(R.median(x)) and float(R.median(x, omitnan=True)) == -3.0
