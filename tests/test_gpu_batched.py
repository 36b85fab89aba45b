"""GPU parity tests for batchinv / batchdet / batchmatvec (general small matrices)."""
import numpy as np
import pytest
import torch
from conftest import TOL, relerr

pytestmark = pytest.mark.gpu
NS = (1, 2, 3, 4, 5, 6, 7, 8, 12, 16)


def B():
    import nitorch_fastmath_amd as N_
    return N_.batched


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', NS)
def test_golden(dev, golden_batched, dn, n):
    g, k = golden_batched, f'{dn}_n{n}_'
    a, v = t(g[k + 'a'], dev), t(g[k + 'v'], dev)
    assert relerr(B().batchinv(a).cpu().numpy(), g[k + 'inv']) <= TOL[dn]
    assert relerr(B().batchdet(a).cpu().numpy(), g[k + 'det']) <= TOL[dn]
    assert relerr(B().batchmatvec(a, v).cpu().numpy(), g[k + 'matvec']) <= TOL[dn]
    if n in (2, 3):
        # TorchScript closed forms with their det perturbation: bit-identical
        assert np.array_equal(B().batchinv(a, perturb=True).cpu().numpy(), g[k + 'inv_ts'])
        assert np.array_equal(B().batchdet(a).cpu().numpy(), g[k + 'det_ts'])


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_golden_rect(dev, golden_batched, dn):
    g = golden_batched
    r = B().batchmatvec(t(g[dn + '_rect_a'], dev), t(g[dn + '_rect_v'], dev))
    assert relerr(r.cpu().numpy(), g[dn + '_rect_matvec']) <= TOL[dn]


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', NS)
@pytest.mark.parametrize('nb', [1, 65, 3001])
def test_vs_oracle(dev, oracle, dn, n, nb):
    if n > 8 and nb > 1000:
        nb = 500
    dtype = np.float32 if dn == 'f32' else np.float64
    rng = np.random.default_rng(n * 1000 + nb)
    a = (rng.standard_normal((nb, n, n)) + 8 * np.eye(n)).astype(dtype)
    v = rng.standard_normal((nb, n)).astype(dtype)
    assert relerr(B().batchinv(t(a, dev)).cpu().numpy(), oracle.batch_inv(a)) <= TOL[dn]
    assert relerr(B().batchdet(t(a, dev)).cpu().numpy(), oracle.batch_det(a)) <= TOL[dn]
    got = B().batchmatvec(t(a, dev), t(v, dev)).cpu().numpy()
    assert np.array_equal(got, oracle.batch_matvec(a, v))


@pytest.mark.parametrize('n', [3, 4, 8])
def test_pivoting_and_layouts(dev, oracle, n):
    """matrices that NEED row exchanges (zero / tiny leading entries), transposed and
    channel-first layouts, broadcast matvec"""
    rng = np.random.default_rng(n)
    nb = 2000
    a = rng.standard_normal((nb, n, n))
    a[:, 0, 0] = 0.0                       # first pivot is useless without pivoting
    a[::3, 1, 1] = 1e-300
    a += 0.0
    ref = oracle.batch_inv(a)
    cond_ok = np.isfinite(ref).all(axis=(1, 2)) & (np.abs(ref).max(axis=(1, 2)) < 1e6)
    got = B().batchinv(t(a, dev)).cpu().numpy()
    assert relerr(got[cond_ok], ref[cond_ok]) <= 1e-9
    eye = np.einsum('bij,bjk->bik', a[cond_ok], got[cond_ok])
    assert np.abs(eye - np.eye(n)).max() < 1e-8
    # transposed view (row/col strides swapped) and channel-first storage
    at = t(a.transpose(0, 2, 1).copy(), dev).transpose(-1, -2)
    assert relerr(B().batchinv(at).cpu().numpy()[cond_ok], ref[cond_ok]) <= 1e-9
    acf = t(a, dev).reshape(40, 50, n, n).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    assert relerr(B().batchinv(acf).cpu().numpy().reshape(nb, n, n)[cond_ok], ref[cond_ok]) <= 1e-9
    assert relerr(B().batchdet(acf).cpu().numpy().reshape(nb), oracle.batch_det(a)) <= 1e-10
    # broadcast: one matrix, many vectors
    v = rng.standard_normal((nb, n))
    r = B().batchmatvec(t(a[0], dev), t(v, dev)).cpu().numpy()
    assert np.array_equal(r, oracle.batch_matvec(np.broadcast_to(a[0], a.shape), v))


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('n', [2, 3, 4, 5, 6, 7, 8])
def test_transposed_and_interleaved_records(dev, oracle, dn, n):
    """matrices stored transposed (a.mT of a contiguous tensor: packed accesses + a renaming of
    registers, MODE_PACKEDT) and records whose elements are two apart (the real parts of a complex
    field: the covering span is fetched packed, MODE_PACKED2), also at a batch stride and with a tail"""
    dtype = np.float32 if dn == 'f32' else np.float64
    rng = np.random.default_rng(200 + n)
    nb = 2 * 257 + 3
    a = (rng.standard_normal((nb, n, n)) + 6 * np.eye(n)).astype(dtype)
    v = rng.standard_normal((nb, n)).astype(dtype)
    rinv, rdet, rmv = oracle.batch_inv(a), oracle.batch_det(a), oracle.batch_matvec(a, v)
    at = t(a.transpose(0, 2, 1), dev).transpose(-1, -2)                 # strides (n*n, 1, n)
    assert at.stride()[-2:] == (1, n) or n == 1
    inv = B().batchinv(at)
    assert relerr(inv.cpu().numpy(), rinv) <= TOL[dn]
    assert relerr(B().batchdet(at).cpu().numpy(), rdet) <= TOL[dn]
    assert np.array_equal(B().batchmatvec(at, t(v, dev)).cpu().numpy(), rmv)
    assert relerr(B().batchinv(at[::2]).cpu().numpy(), rinv[::2]) <= TOL[dn]
    z = torch.full((nb, n, n, 2), float('nan'), dtype=at.dtype, device=dev)
    z[..., 0] = t(a, dev)
    zv = torch.full((nb, n, 2), float('nan'), dtype=at.dtype, device=dev)
    zv[..., 1] = t(v, dev)
    assert relerr(B().batchinv(z[..., 0]).cpu().numpy(), rinv) <= TOL[dn]
    assert relerr(B().batchdet(z[..., 0]).cpu().numpy(), rdet) <= TOL[dn]
    assert np.array_equal(B().batchmatvec(z[..., 0], zv[..., 1]).cpu().numpy(), rmv)
    assert np.array_equal(B().batchmatvec(z[::3, ..., 0], zv[::3, ..., 1]).cpu().numpy(), rmv[::3])


@pytest.mark.parametrize('n', [2, 3, 4, 8])
def test_matrix_first_storage_tiles(dev, oracle, n):
    """(n, n, B, S) storage viewed as (B, S, n, n): component runs -> SoA tile path; S = 340 keeps
    the runs 16-byte aligned (256-lane tiles), S = 341 / 1031 does not (shifted images, and the
    512-lane variant where the image fits)"""
    for Bn, Sn in ((3, 340), (3, 341), (2, 1031), (1, 7)):
        _matrix_first_case(dev, oracle, n, Bn, Sn)


def _matrix_first_case(dev, oracle, n, Bn, Sn):
    rng = np.random.default_rng(40 + n + Sn)
    a = rng.standard_normal((Bn, Sn, n, n)) + 6 * np.eye(n)
    ad = t(a, dev).permute(2, 3, 0, 1).contiguous().permute(2, 3, 0, 1)
    assert not ad.is_contiguous()
    assert relerr(B().batchinv(ad).cpu().numpy(), oracle.batch_inv(a)) <= 1e-12
    assert relerr(B().batchdet(ad).cpu().numpy(), oracle.batch_det(a)) <= 1e-12
    v = rng.standard_normal((Bn, Sn, n))
    vd = t(v, dev).permute(2, 0, 1).contiguous().permute(1, 2, 0)
    assert np.array_equal(B().batchmatvec(ad, vd).cpu().numpy(), oracle.batch_matvec(a, v))


def test_large_inverse_roundtrip(dev):
    """C3-shaped property check: inv(inv(A)) == A and A inv(A) == I on 1e6 8x8 fp64"""
    n = 1_000_000
    g = torch.Generator(device=dev).manual_seed(3)
    a = torch.randn(n, 8, 8, device=dev, dtype=torch.float64, generator=g) + 8 * torch.eye(8, device=dev, dtype=torch.float64)
    ia = B().batchinv(a)
    assert torch.equal(ia, B().batchinv(a))
    back = B().batchinv(ia)
    assert ((back - a).abs().amax() / a.abs().amax()).item() < 1e-12
    v = torch.randn(n, 8, device=dev, dtype=torch.float64, generator=g)
    w = B().batchmatvec(ia, B().batchmatvec(a, v))
    assert ((w - v).abs().amax() / v.abs().amax()).item() < 1e-12
    d = B().batchdet(a) * B().batchdet(ia)
    assert (d - 1).abs().amax().item() < 1e-11


def test_empty(dev):
    assert B().batchinv(torch.zeros(0, 4, 4, device=dev)).shape == (0, 4, 4)
    assert B().batchdet(torch.zeros(2, 0, 3, 3, device=dev)).shape == (2, 0)
