"""GPU parity tests (through the C ABI) for the compact-symmetric family.
Oracle = CPU restatement pinned to the reference's golden vectors; tolerances are
BASELINE.json's: 1e-6 rel fp32 / 1e-12 rel fp64 (max-norm), bit-exact where the
closed forms apply (M <= 4)."""
import numpy as np
import pytest
import torch
from conftest import TOL, relerr

pytestmark = pytest.mark.gpu
MS = (1, 2, 3, 4, 5, 6, 7, 8, 12, 16)
DT = {'f32': torch.float32, 'f64': torch.float64}


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
    c = np.concatenate([np.stack([A[:, i, i] for i in range(M)], -1)] +
                       ([np.stack([A[:, i, j] for i, j in iu], -1)] if iu else []), -1)
    return c.astype(dtype), rng.standard_normal((n, M)).astype(dtype)


def check(got, ref, dn, exact):
    got = got.cpu().numpy()
    assert got.shape == ref.shape and got.dtype == ref.dtype
    if exact:
        assert np.array_equal(got, ref), relerr(got, ref)
    else:
        assert relerr(got, ref) <= TOL[dn], relerr(got, ref)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', MS)
def test_golden_all_ops(dev, golden_sym, dn, M):
    """committed fixtures from the real reference"""
    g, k = golden_sym, f'{dn}_M{M}_'
    mat, vec, inp = t(g[k + 'mat'], dev), t(g[k + 'vec'], dev), t(g[k + 'inp'], dev)
    S = N().sym
    ex = M <= 4
    check(S.sym_solve(mat, vec), g[k + 'solve'], dn, ex)
    check(S.sym_matvec(mat, vec), g[k + 'matvec'], dn, True)
    check(S.sym_addmatvec(inp, mat, vec), g[k + 'addmatvec'], dn, True)
    check(S.sym_submatvec(inp, mat, vec), g[k + 'submatvec'], dn, True)
    check(S.sym_invert(mat), g[k + 'invert'], dn, ex)
    check(S.sym_invert(mat, diag=True), g[k + 'invert_diag'], dn, ex)
    check(S.sym_det(mat), g[k + 'det'], dn, ex)
    check(S.sym_to_full(mat), g[k + 'to_full'], dn, True)
    check(S.sym_outer(vec), g[k + 'outer'], dn, True)
    if M >= 2:
        imat = t(g[k + 'mat_indef'], dev)
        check(S.sym_solve(imat, vec), g[k + 'solve_indef'], dn, ex)
        check(S.sym_invert(imat), g[k + 'invert_indef'], dn, ex)
        check(S.sym_solve(mat[:, :M].contiguous(), vec), g[k + 'solve_diag'], dn, True)
        check(S.sym_matvec(mat[:, :M].contiguous(), vec), g[k + 'matvec_diag'], dn, True)
        check(S.sym_solve(mat[:, :1].contiguous(), vec), g[k + 'solve_scal'], dn, True)
        check(S.sym_matvec(mat[:, :1].contiguous(), vec), g[k + 'matvec_scal'], dn, True)
    if M > 2:
        full = t(g[k + 'to_full'].reshape(len(g[k + 'mat']), M * M), dev)
        check(S.sym_solve(full, vec), g[k + 'solve'], dn, False)
        check(S.sym_matvec(full, vec), g[k + 'matvec'], dn, False)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('kd', [(1, 1), (2, 2), (3, 3), (3, 2), (4, 4), (2, 3)])
def test_golden_matmul(dev, golden_sym, dn, kd):
    g, k = golden_sym, f'{dn}_k{kd[0]}_d{kd[1]}_'
    check(N().sym.sym_matmul(t(g[k + 'j'], dev), t(g[k + 'h'], dev)), g[k + 'matmul'], dn, True)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', MS)
@pytest.mark.parametrize('n', [1, 63, 257, 5000])
def test_vs_oracle_sizes(dev, oracle, dn, M, n):
    """seeded inputs, ragged sizes (tile tails), tiled AoS path"""
    if M > 8 and n > 1000:
        n = 700
    mat, vec = spd_np(n, M, np.float32 if dn == 'f32' else np.float64, 100 * M + n)
    S = N().sym
    ex = M <= 4
    check(S.sym_solve(t(mat, dev), t(vec, dev)), oracle.sym_solve(mat, vec), dn, ex)
    check(S.sym_matvec(t(mat, dev), t(vec, dev)), oracle.sym_matvec(mat, vec), dn, True)
    check(S.sym_invert(t(mat, dev)), oracle.sym_invert(mat), dn, ex)
    check(S.sym_invert(t(mat, dev), diag=True), oracle.sym_invert(mat, diag=True), dn, ex)
    check(S.sym_det(t(mat, dev)), oracle.sym_det(mat), dn, ex)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', [2, 3, 4, 6, 8, 12])
def test_layouts(dev, oracle, dn, M):
    """channel-first (SoA) fields, two-level batches, broadcast, misaligned views, out="""
    dtype = np.float32 if dn == 'f32' else np.float64
    K = M * (M + 1) // 2
    B, X, Y = 3, 17, 9
    mat, vec = spd_np(B * X * Y, M, dtype, 7 + M)
    ref = oracle.sym_solve(mat, vec).reshape(B, X, Y, M)
    refmv = oracle.sym_matvec(mat, vec).reshape(B, X, Y, M)
    ex = M <= 4
    S = N().sym
    mat4, vec4 = t(mat, dev).reshape(B, X, Y, K), t(vec, dev).reshape(B, X, Y, M)
    # (B, C, X, Y) channel-first storage viewed channel-last: no copy, two batch levels
    mat_cf = mat4.movedim(-1, 1).contiguous().movedim(1, -1)
    vec_cf = vec4.movedim(-1, 1).contiguous().movedim(1, -1)
    assert not mat_cf.is_contiguous()
    check(S.sym_solve(mat_cf, vec_cf), ref, dn, ex)
    check(S.sym_matvec(mat_cf, vec4), refmv, dn, True)          # mixed layouts
    # channel-first output buffer through out=
    out_cf = torch.empty(B, M, X, Y, dtype=vec4.dtype, device=dev).movedim(1, -1)
    r = S.sym_solve(mat_cf, vec_cf, out=out_cf)
    assert r.data_ptr() == out_cf.data_ptr()
    check(out_cf, ref, dn, ex)
    # pure SoA: (K, n).T
    mat_soa = t(mat, dev).t().contiguous().t()
    vec_soa = t(vec, dev).t().contiguous().t()
    check(S.sym_solve(mat_soa, vec_soa), ref.reshape(-1, M), dn, ex)
    # broadcast: one matrix against many vectors, and (1, X, 1, K) against (B, X, Y, M)
    ref_b = oracle.sym_solve(np.broadcast_to(mat[:1], (len(vec), K)), vec)
    check(S.sym_solve(t(mat[0], dev), t(vec, dev)), ref_b, dn, ex)
    mp = mat.reshape(B, X, Y, K)[:1, :, :1]
    ref_p = oracle.sym_solve(np.broadcast_to(mp, (B, X, Y, K)), vec.reshape(B, X, Y, M))
    check(S.sym_solve(t(mp, dev), vec4), ref_p, dn, ex)
    # misaligned base pointer (slice off one record) and strided batch
    check(S.sym_solve(t(mat, dev)[1:], t(vec, dev)[1:]), ref.reshape(-1, M)[1:], dn, ex)
    check(S.sym_solve(t(mat, dev)[::2], t(vec, dev)[::2]), ref.reshape(-1, M)[::2], dn, ex)
    # a view with three genuine stride levels (materialised by the facade)
    big = torch.zeros(B, 2, X, 2, Y, K, dtype=mat4.dtype, device=dev)
    big[:, 0, :, 0] = mat4
    check(S.sym_solve(big[:, 0, :, 0], vec4), ref, dn, ex)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', [2, 3, 4, 5, 6, 7])
def test_packed_records_at_any_batch_stride(dev, oracle, dn, M):
    """records that are contiguous inside but not back to back (every k-th record, padded records,
    a padded output buffer) move with whole 16-byte accesses per lane (MODE_PACKED): same results,
    and the gaps of a padded output are left alone"""
    dtype = np.float32 if dn == 'f32' else np.float64
    K = M * (M + 1) // 2
    n = 3 * 257 + 5
    mat, vec = spd_np(n, M, dtype, 90 + M)
    ref = oracle.sym_solve(mat, vec)
    refmv = oracle.sym_matvec(mat, vec)
    refinv = oracle.sym_invert(mat)
    ex = M <= 4
    S = N().sym
    md, vd = t(mat, dev), t(vec, dev)
    for k in (2, 3):
        check(S.sym_solve(md[::k], vd[::k]), ref[::k], dn, ex)
        check(S.sym_matvec(md[::k], vd[::k].contiguous()), refmv[::k], dn, True)
        check(S.sym_invert(md[::k]), refinv[::k], dn, ex)
    # padded records: (n, K + 3)[:, :K] and (n, M + 1)[:, 1:] (the latter starts off a 16-byte line)
    mp = torch.full((n, K + 3), float('nan'), dtype=md.dtype, device=dev)
    mp[:, :K] = md
    vp = torch.full((n, M + 1), float('nan'), dtype=md.dtype, device=dev)
    vp[:, 1:] = vd
    check(S.sym_solve(mp[:, :K], vp[:, 1:]), ref, dn, ex)
    # padded output buffers: only the records are written
    op = torch.full((n, M + 2), 7.0, dtype=md.dtype, device=dev)
    r = S.sym_solve(md, vd, out=op[:, 1:M + 1])
    assert r.data_ptr() == op[:, 1:M + 1].data_ptr()
    check(op[:, 1:M + 1], ref, dn, ex)
    assert bool((op[:, 0] == 7).all()) and bool((op[:, M + 1] == 7).all())
    ip = torch.full((n, K + 1), 7.0, dtype=md.dtype, device=dev)
    S.sym_invert(mp[:, :K], out=ip[:, :K])
    check(ip[:, :K], refinv, dn, ex)
    assert bool((ip[:, K] == 7).all())
    # elements two apart (one of two interleaved fields): covering span fetched packed (inputs);
    # an interleaved OUTPUT is written element by element, the other field is left alone
    z = torch.full((n, K, 2), float('nan'), dtype=md.dtype, device=dev)
    z[..., 1] = md
    zv = torch.full((n, M, 2), 7.0, dtype=md.dtype, device=dev)
    zv[..., 0] = vd
    check(S.sym_solve(z[..., 1], zv[..., 0]), ref, dn, ex)
    check(S.sym_solve(z[::2, :, 1], zv[::2, :, 0]), ref[::2], dn, ex)
    S.sym_solve(z[..., 1], zv[..., 0].clone(), out=zv[..., 0])
    check(zv[..., 0], ref, dn, ex)
    assert bool((zv[..., 1] == 7).all())
    # in place on strided views: every lane reads its record before it writes it
    vb = torch.zeros(2 * n, M, dtype=md.dtype, device=dev)
    vb[::2] = vd
    r = S.sym_solve_(md, vb[::2])
    assert r.data_ptr() == vb.data_ptr()
    check(vb[::2], ref, dn, ex)
    assert bool((vb[1::2] == 0).all())
    mb = torch.zeros(n, 2 * K, dtype=md.dtype, device=dev)
    mb[:, :K] = md
    S.sym_invert_(mb[:, :K])
    check(mb[:, :K], refinv, dn, ex)
    assert bool((mb[:, K:] == 0).all())
    # cropped 2-D field: rows of contiguous records, outer stride larger than the row
    X, Y = 3, 257
    m2, v2 = md[:X * Y].reshape(X, Y, K), vd[:X * Y].reshape(X, Y, M)
    check(S.sym_solve(m2[:, 3:200], v2[:, 3:200]), ref[:X * Y].reshape(X, Y, M)[:, 3:200], dn, ex)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', [3, 4, 6, 12])
def test_inplace_and_eps_and_dtype(dev, oracle, dn, M):
    dtype = np.float32 if dn == 'f32' else np.float64
    n = 1000
    mat, vec = spd_np(n, M, dtype, 31 + M)
    inp = np.random.default_rng(5).standard_normal((n, M)).astype(dtype)
    S = N().sym
    ex = M <= 4
    v = t(vec, dev)
    r = S.sym_solve_(t(mat, dev), v)
    assert r.data_ptr() == v.data_ptr()
    check(v, oracle.sym_solve(mat, vec), dn, ex)
    m = t(mat, dev)
    r = S.sym_invert_(m)
    assert r.data_ptr() == m.data_ptr()
    check(m, oracle.sym_invert(mat), dn, ex)
    i = t(inp, dev)
    S.sym_addmatvec_(i, t(mat, dev), t(vec, dev))
    check(i, oracle.sym_matvec(mat, vec, inp, +1), dn, True)
    i = t(inp, dev)
    S.sym_submatvec_(i, t(mat, dev), t(vec, dev))
    check(i, oracle.sym_matvec(mat, vec, inp, -1), dn, True)
    # eps: documented intent = add to the diagonal (last value repeated)
    eps = [0.5, 0.25]
    e = np.array((eps + [eps[-1]] * M)[:M], dtype)
    mat_e = mat.copy()
    mat_e[:, :M] += e
    check(S.sym_solve(t(mat, dev), t(vec, dev), eps=eps), oracle.sym_solve(mat_e, vec), dn, ex)
    mat_s = mat.copy()
    mat_s[:, :M] += dtype(0.125)
    check(S.sym_solve(t(mat, dev), t(vec, dev), eps=0.125), oracle.sym_solve(mat_s, vec), dn, ex)
    # dtype= : computation dtype
    if dn == 'f32':
        r = S.sym_solve(t(mat, dev), t(vec, dev), dtype=torch.float64)
        assert r.dtype == torch.float64
        check(r, oracle.sym_solve(mat.astype(np.float64), vec.astype(np.float64)), 'f64', ex)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', [2, 4, 6, 8])
def test_component_major_tiles(dev, oracle, dn, M):
    """channel-first fields take the SoA tile path (MODE_SOA): aligned vector loads of every
    component run through LDS; sizes with full and partial tiles, voxel counts that are and
    are not multiples of 4 (odd volumes: every component run starts at another alignment)"""
    dtype = np.float32 if dn == 'f32' else np.float64
    K = M * (M + 1) // 2
    S = N().sym
    ex = M <= 4
    for B, X, Y in ((2, 40, 20), (1, 4, 3), (3, 16, 17), (2, 7, 9), (1, 33, 31), (3, 5, 5), (1, 1, 3), (2, 1, 1)):
        mat, vec = spd_np(B * X * Y, M, dtype, 77 + M + X)
        ref = oracle.sym_solve(mat, vec).reshape(B, X, Y, M)
        refi = oracle.sym_invert(mat).reshape(B, X, Y, K)
        refmv = oracle.sym_matvec(mat, vec, vec, -1).reshape(B, X, Y, M)
        mat_cf = t(mat, dev).reshape(B, X, Y, K).movedim(-1, 1).contiguous().movedim(1, -1)
        vec_cf = t(vec, dev).reshape(B, X, Y, M).movedim(-1, 1).contiguous().movedim(1, -1)
        out_cf = torch.empty(B, M, X, Y, dtype=vec_cf.dtype, device=dev).movedim(1, -1)
        S.sym_solve(mat_cf, vec_cf, out=out_cf)
        check(out_cf, ref, dn, ex)
        check(S.sym_solve(mat_cf, vec_cf), ref, dn, ex)                 # SoA in, AoS out
        check(S.sym_solve(mat_cf, vec_cf.contiguous()), ref, dn, ex)    # mixed inputs
        inv_cf = torch.empty(B, K, X, Y, dtype=vec_cf.dtype, device=dev).movedim(1, -1)
        S.sym_invert(mat_cf, out=inv_cf)
        check(inv_cf, refi, dn, ex)
        check(S.sym_submatvec(vec_cf, mat_cf, vec_cf), refmv, dn, True)
        # a window of a larger channel-first buffer: runs start and end inside 16-byte vectors,
        # and whatever surrounds the window must stay untouched
        n = B * X * Y
        if n >= 8:
            pad = torch.full((M, n + 7), 123.0, dtype=vec_cf.dtype, device=dev)
            win = pad[:, 3:3 + n].t()
            S.sym_solve(mat_cf.reshape(n, K), vec_cf.reshape(n, M), out=win)
            check(win, ref.reshape(n, M), dn, ex)
            assert bool((pad[:, :3] == 123.0).all()) and bool((pad[:, 3 + n:] == 123.0).all())
        v2 = vec_cf.clone(memory_format=torch.preserve_format)
        S.sym_solve_(mat_cf, v2)                                        # in place on an SoA buffer
        check(v2, ref, dn, ex)
    # pure (K, n).T with n a multiple of 4 and not of the tile
    n = 1000 + 4
    mat, vec = spd_np(n, M, dtype, 5 + M)
    ms, vs = t(mat, dev).t().contiguous().t(), t(vec, dev).t().contiguous().t()
    check(S.sym_solve(ms, vs), oracle.sym_solve(mat, vec), dn, ex)


def test_empty_and_errors(dev):
    S = N().sym
    r = S.sym_solve(torch.zeros(0, 10, device=dev), torch.zeros(0, 4, device=dev))
    assert r.shape == (0, 4)
    assert S.sym_invert(torch.zeros(3, 0, 6, device=dev)).shape == (3, 0, 6)
    with pytest.raises(ValueError):
        S.sym_solve(torch.zeros(5, 7, device=dev), torch.zeros(5, 4, device=dev))
    with pytest.raises(ValueError):
        S.sym_invert(torch.zeros(5, 7, device=dev))
    with pytest.raises(TypeError):
        S.sym_solve(torch.zeros(5, 10, device=dev, dtype=torch.half), torch.zeros(5, 4, device=dev, dtype=torch.half))
    # order 17 (K = 153): beyond the kernels, served by torch.linalg on the device like upstream
    r17 = S.sym_solve(torch.eye(17, device=dev)[None].expand(5, 17, 17)[..., :1].new_ones(5, 153) * 0 +
                      torch.cat([torch.ones(5, 17, device=dev), torch.zeros(5, 136, device=dev)], -1),
                      torch.arange(17., device=dev).expand(5, 17))
    assert torch.equal(r17, torch.arange(17., device=dev).expand(5, 17))
    # singular input -> inf/nan like the reference, no error
    r = S.sym_solve(torch.zeros(4, 10, device=dev), torch.ones(4, 4, device=dev))
    assert not torch.isfinite(r).any()


@pytest.mark.parametrize('M,n', [(4, 20_000_000), (6, 10_000_000), (3, 5_000_000)])
def test_large_roundtrip_properties(dev, M, n):
    """full-size style checks that need no oracle: solve(A, A v) == v, A inv(A) == I
    (through matvec), bit-reproducibility, and sampled agreement with the oracle."""
    import oracle as O
    S = N().sym
    from bench import spd_compact     # element-wise generator (no batched GEMM)
    mat, v = spd_compact(n, M, torch.float32, dev, 99)
    y = S.sym_matvec(mat, v)
    x = S.sym_solve(mat, y)
    err = ((x - v).abs().amax() / v.abs().amax()).item()
    assert err < 2e-5, err     # cond <= ~10, fp32
    x2 = S.sym_solve(mat, y)
    assert torch.equal(x, x2)
    # linearity: solve(A, 2y) == 2 solve(A, y) exactly (power-of-two scaling)
    assert torch.equal(S.sym_solve(mat, 2 * y), 2 * x)
    # sampled oracle agreement at scattered offsets, incl. the last (partial) tile
    idx = torch.cat([torch.arange(0, 4096), torch.arange(n // 2, n // 2 + 4096), torch.arange(n - 777, n)]).to(dev)
    ref = O.sym_solve(mat[idx].cpu().numpy(), y[idx].cpu().numpy())
    got = x[idx].cpu().numpy()
    if M <= 4:
        assert np.array_equal(got, ref)
    else:
        assert relerr(got, ref) <= 1e-6
    inv = S.sym_invert(mat)
    e = S.sym_matvec(inv, y)            # inv(A) (A v) == v
    assert ((e - v).abs().amax() / v.abs().amax()).item() < 2e-5


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', [2, 3, 4, 5, 6, 8])
def test_odd_row_offsets_of_contiguous_tensors(dev, oracle, dn, M):
    """slices `x[i0:]` of contiguous (n, K) tensors start at any multiple of the record size:
    the launcher peels 1..3 records off so that the rest runs through the aligned tile path
    (or falls back to per-lane loads when no peel aligns every operand) -- same results"""
    dtype = np.float32 if dn == 'f32' else np.float64
    S = N().sym
    ex = M <= 4
    n = 3000
    mat, vec = spd_np(n, M, dtype, 300 + M)
    md, vd = t(mat, dev), t(vec, dev)
    for i0 in (1, 2, 3, 5):
        for i1 in (n, n - 1, i0 + 1, i0 + 2, i0 + 300):
            ref = oracle.sym_solve(mat[i0:i1], vec[i0:i1])
            check(S.sym_solve(md[i0:i1], vd[i0:i1]), ref, dn, ex)
            out = torch.full((n, M), 7.0, dtype=vd.dtype, device=dev)
            S.sym_solve(md[i0:i1], vd[i0:i1], out=out[i0:i1])
            check(out[i0:i1], ref, dn, ex)
            assert bool((out[:i0] == 7).all()) and bool((out[i1:] == 7).all())
            v2 = vd.clone()
            S.sym_solve_(md[i0:i1], v2[i0:i1])                       # in place on a slice
            check(v2[i0:i1], ref, dn, ex)
            assert torch.equal(v2[:i0], vd[:i0]) and torch.equal(v2[i1:], vd[i1:])
        check(S.sym_invert(md[i0:]), oracle.sym_invert(mat[i0:]), dn, ex)
        check(S.sym_matvec(md[i0:], vd[i0:]), oracle.sym_matvec(mat[i0:], vec[i0:]), dn, True)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_fused_matmul_solve_equals_the_chain(dev, oracle, dn):
    """EXTENSION `sym_matmul_solve`: one kernel, bit-identical to sym_solve(sym_matmul(j, h), g, eps)
    (and therefore to the oracle chain wherever the chain is)"""
    dtype = np.float32 if dn == 'f32' else np.float64
    S = N().sym
    rng = np.random.default_rng(17)

    def same(x, y):       # k < d makes J^T H J singular: inf / nan must match too
        return bool(((x == y) | (torch.isnan(x) & torch.isnan(y))).all())
    n = 1237
    for k in (1, 2, 3, 4):
        for d in (1, 2, 3, 4):
            j = rng.standard_normal((n, k, d)).astype(dtype) + (np.eye(k, d, dtype=dtype) * 2)
            h, _ = spd_np(n, k, dtype, 500 + 10 * k + d)
            g = rng.standard_normal((n, d)).astype(dtype)
            jd, hd, gd = t(j, dev), t(h, dev), t(g, dev)
            for eps in (None, 0.25, [0.5, 0.125]):
                chain = S.sym_solve(S.sym_matmul(jd, hd), gd, eps=eps)
                fused = S.sym_matmul_solve(jd, hd, gd, eps=eps)
                assert same(fused, chain), (k, d, eps)
            a = oracle.sym_matmul(j, h)
            ref = oracle.sym_solve(a, g)
            assert np.array_equal(S.sym_matmul_solve(jd, hd, gd).cpu().numpy(), ref, equal_nan=True), (k, d)
            # diagonal hessian, channel-first operands, one hessian for all jacobians
            hdiag = hd[:, :k].contiguous()
            assert same(S.sym_matmul_solve(jd, hdiag, gd), S.sym_solve(S.sym_matmul(jd, hdiag), gd))
            jc = jd.permute(1, 2, 0).contiguous().permute(2, 0, 1)
            hc, gc = hd.t().contiguous().t(), gd.t().contiguous().t()
            r = S.sym_matmul_solve(jc, hc, gc)
            assert same(r, S.sym_solve(S.sym_matmul(jd, hd), gd))
            if d > 1:
                assert r.stride(-1) != 1          # channel-first in, channel-first out
            assert same(S.sym_matmul_solve(jd, hd[:1], gd), S.sym_solve(S.sym_matmul(jd, hd[:1]), gd))
    # sizes beyond 4 run the chain; gradients flow through the chain's autograd
    torch.manual_seed(77)
    j = torch.randn(50, 5, 3, device=dev, dtype=torch.float64)
    h, _ = spd_np(50, 5, np.float64, 1)
    g = torch.randn(50, 3, device=dev, dtype=torch.float64)
    assert torch.equal(S.sym_matmul_solve(j, t(h, dev), g), S.sym_solve(S.sym_matmul(j, t(h, dev)), g))
    jr = torch.randn(9, 3, 3, device=dev, dtype=torch.float64, requires_grad=True)
    S.sym_matmul_solve(jr, t(spd_np(9, 3, np.float64, 2)[0], dev), torch.randn(9, 3, device=dev, dtype=torch.float64)).sum().backward()
    assert jr.grad is not None and bool(torch.isfinite(jr.grad).all())


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('M', [2, 3, 4, 5, 6, 8])
def test_one_matrix_many_vectors(dev, oracle, dn, M):
    """a matrix that is the same along the inner batch level (stride 0) takes sym_solve_bcast_kernel:
    cofactors once per lane, V right-hand sides each -- the same operations in the same order as the
    per-record kernel, so still bit-identical to the oracle (closed forms)"""
    dtype = np.float32 if dn == 'f32' else np.float64
    K = M * (M + 1) // 2
    B, n = 3, 4 * 1024 + 37                        # ragged tail, several workgroups
    mat, _ = spd_np(B, M, dtype, 50 + M)
    vec = np.random.default_rng(60 + M).standard_normal((B, n, M)).astype(dtype)
    S = N().sym
    md, vd = t(mat, dev), t(vec, dev)
    full = np.broadcast_to(mat[:, None, :], (B, n, K))
    ref = oracle.sym_solve(np.ascontiguousarray(full).reshape(-1, K), vec.reshape(-1, M)).reshape(B, n, M)
    ex = M <= 4                                                              # closed forms: bit-identical to the oracle
    check(S.sym_solve(md[0], vd[0]), ref[0], dn, ex)                         # (K,) against (n, M)
    check(S.sym_solve(md[:, None, :], vd), ref, dn, ex)                      # (B, 1, K) against (B, n, M)
    check(S.sym_solve(md[1], vd[1, ::2]), ref[1, ::2], dn, ex)               # strided vectors
    # a materialised copy of the same matrix goes through the per-record kernel: the same bits at every order
    per_record = S.sym_solve(md[0].expand(n, K).contiguous(), vd[0])
    check(per_record, ref[0], dn, ex)
    assert torch.equal(S.sym_solve(md[0], vd[0]), per_record)
    # out= into a padded buffer, in place, and eps on the diagonal
    buf = torch.full((n, M + 1), 7.0, dtype=md.dtype, device=dev)
    S.sym_solve(md[2], vd[2], out=buf[:, :M])
    check(buf[:, :M], ref[2], dn, ex)
    assert bool((buf[:, M] == 7).all())
    v = vd[0].clone()
    S.sym_solve_(md[0], v)
    check(v, ref[0], dn, ex)
    eps = [0.5, 0.25]
    e = np.array((eps + [eps[-1]] * M)[:M], dtype)
    mat_e = mat.copy()
    mat_e[:, :M] += e
    ref_e = oracle.sym_solve(np.ascontiguousarray(np.broadcast_to(mat_e[0], (n, K))), vec[0])
    check(S.sym_solve(md[0], vd[0], eps=eps), ref_e, dn, ex)
