"""ctypes front end of oracle/libnfm_oracle.so (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY -- the checker, never the thing measured or shipped.
All arrays are made C-contiguous batch-major before the call; batch dims are
flattened.  See nfm_oracle.c for the reference file:line each routine follows.
"""
import ctypes
import os
import subprocess
import numpy as np

__all__ = ['build', 'lib', 'sym_solve', 'sym_matvec', 'sym_invert', 'sym_det', 'sym_to_full',
           'sym_outer', 'sym_matmul', 'batch_inv', 'batch_det', 'batch_matvec', 'reduce', 'givens', 'givens_apply', 'householder',
           'householder_apply', 'hessenberg', 'hessenberg_sym', 'qr_hessenberg', 'rq_hessenberg', 'eig_sym', 'median',
           'set_num_threads']

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libnfm_oracle.so')
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = [os.path.join(_HERE, f) for f in ('nfm_oracle.c', 'nfm_oracle_body.inc', 'nfm_oracle_qr.inc')]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.run(['make', '-C', _HERE, '-B'], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def set_num_threads(n):
    """OpenMP thread count used by the batch loops (cpu_baseline reports it)."""
    omp = ctypes.CDLL('libgomp.so.1')
    omp.omp_set_num_threads(int(n))


def _dt(a):
    if a.dtype == np.float32:
        return 0
    if a.dtype == np.float64:
        return 1
    raise TypeError(a.dtype)


def _c(a, dtype=None):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _chk(rc):
    if rc != 0:
        raise RuntimeError(f'nfm_oracle error {rc}')


_KIND = {'sym': 0, 'diag': 1, 'scal': 2, 'full': 3}


def _mat_kind(NN, M):
    # sym.py:16-24 -- NN in {1, N, N(N+1)/2, N*N}; ambiguous sizes resolve in this order
    if NN == M * (M + 1) // 2:
        return 0
    if NN == M:
        return 1
    if NN == 1:
        return 2
    if NN == M * M:
        return 3
    raise ValueError((NN, M))


def sym_solve(mat, vec):
    mat, vec = _c(mat), _c(vec, mat.dtype)
    M = vec.shape[-1]
    batch = np.broadcast_shapes(mat.shape[:-1], vec.shape[:-1])
    mat = _c(np.broadcast_to(mat, batch + mat.shape[-1:]))
    vec = _c(np.broadcast_to(vec, batch + (M,)))
    out = np.empty_like(vec)
    n = int(np.prod(batch, dtype=np.int64))
    _chk(lib().nfm_oracle_sym_solve(_dt(mat), M, ctypes.c_int64(n), _mat_kind(mat.shape[-1], M),
                                    _p(mat), _p(vec), _p(out)))
    return out


def sym_matvec(mat, vec, inp=None, sign=1):
    mat, vec = _c(mat), _c(vec, mat.dtype)
    M = vec.shape[-1]
    shapes = [mat.shape[:-1], vec.shape[:-1]] + ([inp.shape[:-1]] if inp is not None else [])
    batch = np.broadcast_shapes(*shapes)
    mat = _c(np.broadcast_to(mat, batch + mat.shape[-1:]))
    vec = _c(np.broadcast_to(vec, batch + (M,)))
    if inp is not None:
        inp = _c(np.broadcast_to(_c(inp, mat.dtype), batch + (M,)))
    out = np.empty_like(vec)
    n = int(np.prod(batch, dtype=np.int64))
    _chk(lib().nfm_oracle_sym_matvec(_dt(mat), M, ctypes.c_int64(n), _mat_kind(mat.shape[-1], M), sign,
                                     _p(mat), _p(vec), _p(inp), _p(out)))
    return out


def _M_of(K):
    M = int((np.sqrt(1 + 8 * K) - 1) // 2)
    assert M * (M + 1) // 2 == K, K
    return M


def sym_invert(mat, diag=False):
    mat = _c(mat)
    K = mat.shape[-1]
    M = _M_of(K)
    out = np.empty(mat.shape[:-1] + ((M if diag else K),), mat.dtype)
    n = int(np.prod(mat.shape[:-1], dtype=np.int64))
    _chk(lib().nfm_oracle_sym_invert(_dt(mat), M, ctypes.c_int64(n), int(diag), _p(mat), _p(out)))
    return out


def sym_det(mat):
    mat = _c(mat)
    M = _M_of(mat.shape[-1])
    out = np.empty(mat.shape[:-1], mat.dtype)
    n = int(np.prod(mat.shape[:-1], dtype=np.int64))
    _chk(lib().nfm_oracle_sym_det(_dt(mat), M, ctypes.c_int64(n), _p(mat), _p(out)))
    return out


def sym_to_full(mat):
    mat = _c(mat)
    M = _M_of(mat.shape[-1])
    out = np.empty(mat.shape[:-1] + (M, M), mat.dtype)
    n = int(np.prod(mat.shape[:-1], dtype=np.int64))
    _chk(lib().nfm_oracle_sym_to_full(_dt(mat), M, ctypes.c_int64(n), _p(mat), _p(out)))
    return out


def sym_outer(x):
    x = _c(x)
    M = x.shape[-1]
    out = np.empty(x.shape[:-1] + (M * (M + 1) // 2,), x.dtype)
    n = int(np.prod(x.shape[:-1], dtype=np.int64))
    _chk(lib().nfm_oracle_sym_outer(_dt(x), M, ctypes.c_int64(n), _p(x), _p(out)))
    return out


def sym_matmul(j, h):
    j = _c(j)
    h = _c(h, j.dtype)
    K, D = j.shape[-2:]
    hess_diag = int(h.shape[-1] == K and K != 1)
    batch = np.broadcast_shapes(j.shape[:-2], h.shape[:-1])
    j = _c(np.broadcast_to(j, batch + (K, D)))
    h = _c(np.broadcast_to(h, batch + h.shape[-1:]))
    out = np.empty(batch + (D * (D + 1) // 2,), j.dtype)
    n = int(np.prod(batch, dtype=np.int64))
    _chk(lib().nfm_oracle_sym_matmul(_dt(j), K, D, ctypes.c_int64(n), hess_diag, _p(j), _p(h), _p(out)))
    return out


def batch_inv(a, closed=False):
    a = _c(a)
    N = a.shape[-1]
    out = np.empty_like(a)
    n = int(np.prod(a.shape[:-2], dtype=np.int64))
    _chk(lib().nfm_oracle_batch_inv(_dt(a), N, ctypes.c_int64(n), int(closed), _p(a), _p(out)))
    return out


def batch_det(a, closed=False):
    a = _c(a)
    N = a.shape[-1]
    out = np.empty(a.shape[:-2], a.dtype)
    n = int(np.prod(a.shape[:-2], dtype=np.int64))
    _chk(lib().nfm_oracle_batch_det(_dt(a), N, ctypes.c_int64(n), int(closed), _p(a), _p(out)))
    return out


def batch_matvec(a, v):
    a = _c(a)
    v = _c(v, a.dtype)
    rows, cols = a.shape[-2:]
    batch = np.broadcast_shapes(a.shape[:-2], v.shape[:-1])
    a = _c(np.broadcast_to(a, batch + (rows, cols)))
    v = _c(np.broadcast_to(v, batch + (cols,)))
    out = np.empty(batch + (rows,), a.dtype)
    n = int(np.prod(batch, dtype=np.int64))
    _chk(lib().nfm_oracle_batch_matvec(_dt(a), rows, cols, ctypes.c_int64(n), _p(a), _p(v), _p(out)))
    return out


_OPS = {'nansum': 0, 'nanmax': 1, 'nanmin': 2, 'sum': 3, 'max': 4, 'min': 5}


def reduce(op, x, dim=None, keepdim=False, out_f64=False):
    """Reduce over `dim` (int, sequence or None = all), reference semantics of reduce.py."""
    x = _c(x)
    nd = x.ndim
    if dim is None:
        dims = list(range(nd))
    else:
        dims = [d % nd for d in (dim if isinstance(dim, (list, tuple)) else [dim])]
    keep = [d for d in range(nd) if d not in dims]
    xt = _c(np.transpose(x, keep + dims))
    kept_shape = [x.shape[d] for d in keep]
    outer = int(np.prod(kept_shape, dtype=np.int64))
    red = int(np.prod([x.shape[d] for d in dims], dtype=np.int64))
    out = np.empty(kept_shape, np.float64 if out_f64 else x.dtype)
    _chk(lib().nfm_oracle_reduce(_dt(x), _OPS[op], ctypes.c_int64(outer), ctypes.c_int64(red),
                                 ctypes.c_int64(1), _p(xt), _p(out), int(out_f64)))
    if keepdim:
        out = out.reshape([1 if d in dims else x.shape[d] for d in range(nd)])
    return out


# ------------------------------------------------------------------ QR family
_SIDE = {'left': 0, 'right': 1, 'both': 2}


def _nb(a, nd):
    return int(np.prod(a.shape[:a.ndim - nd], dtype=np.int64))


def givens(x, y):
    x, y = np.broadcast_arrays(_c(x), _c(y))
    x, y = _c(x), _c(y, x.dtype)
    c, s = np.empty_like(x), np.empty_like(x)
    _chk(lib().nfm_oracle_qr_givens(_dt(x), ctypes.c_int64(x.size), _p(x), _p(y), _p(c), _p(s)))
    return c, s


def givens_apply(a, c, s, i=0, j=None, side='both'):
    a = np.array(a, copy=True, order='C')
    N = a.shape[-1]
    j = i + 1 if j is None else j
    shp = a.shape[:-2] + (N,)
    c = _c(np.broadcast_to(np.asarray(c, a.dtype), shp))
    s = _c(np.broadcast_to(np.asarray(s, a.dtype), shp))
    _chk(lib().nfm_oracle_qr_givens_apply(_dt(a), N, ctypes.c_int64(_nb(a, 2)), _SIDE[side], i, j, _p(a), _p(c), _p(s)))
    return a


def householder(x, basis=0):
    x = np.array(x, copy=True, order='C')
    alpha = np.empty(x.shape[:-1], x.dtype)
    _chk(lib().nfm_oracle_qr_householder(_dt(x), x.shape[-1], ctypes.c_int64(_nb(x, 1)), basis, _p(x), _p(alpha)))
    return x, alpha


def householder_apply(a, u, side='both', inverse=False):
    a = np.array(a, copy=True, order='C')
    us = list(u) if isinstance(u, (list, tuple)) else [u]
    if inverse:
        us = us[::-1]
    N = a.shape[-1]
    for uk in us:
        uk = _c(np.broadcast_to(np.asarray(uk, a.dtype), a.shape[:-2] + (uk.shape[-1],)))
        _chk(lib().nfm_oracle_qr_householder_apply(_dt(a), N, uk.shape[-1], ctypes.c_int64(_nb(a, 2)),
                                                   _SIDE[side], _p(a), _p(uk)))
    return a


def _unpack_u(upack, N):
    return [upack[..., k, :N - 1 - k].copy() for k in range(max(N - 2, 0))]


def hessenberg(a, compute_u=False):
    a = np.array(a, copy=True, order='C')
    N = a.shape[-1]
    up = np.zeros(a.shape[:-2] + (max(N - 2, 0), N - 1), a.dtype) if compute_u else None
    _chk(lib().nfm_oracle_qr_hessenberg(_dt(a), N, ctypes.c_int64(_nb(a, 2)), _p(a), _p(up)))
    return (a, _unpack_u(up, N)) if compute_u else a


def hessenberg_sym(a, upper=True, fill=True, compute_u=False):
    a = np.array(a, copy=True, order='C')
    N = a.shape[-1]
    up = np.zeros(a.shape[:-2] + (max(N - 2, 0), N - 1), a.dtype) if compute_u else None
    _chk(lib().nfm_oracle_qr_hessenberg_sym(_dt(a), N, ctypes.c_int64(_nb(a, 2)), int(upper), int(fill), _p(a), _p(up)))
    return (a, _unpack_u(up, N)) if compute_u else a


def qr_hessenberg(h):
    r = np.array(h, copy=True, order='C')
    q = np.empty_like(r)
    _chk(lib().nfm_oracle_qr_qr_hessenberg(_dt(r), r.shape[-1], ctypes.c_int64(_nb(r, 2)), _p(r), _p(q)))
    return q, r


def rq_hessenberg(h, u=None, sym=False, true_rq=True):
    h = np.array(h, copy=True, order='C')
    if u is not None:
        u = np.array(u, dtype=h.dtype, copy=True, order='C')
    _chk(lib().nfm_oracle_qr_rq_hessenberg(_dt(h), h.shape[-1], ctypes.c_int64(_nb(h, 2)), int(sym), int(true_rq),
                                           _p(h), _p(u)))
    return h if u is None else (h, u)


def eig_sym(a, compute_u=False, upper=True, max_iter=1024, tol=1e-32, return_sweeps=False):
    a = np.array(a, copy=True, order='C')
    N = a.shape[-1]
    vals = np.empty(a.shape[:-1], a.dtype)
    vecs = np.empty_like(a) if compute_u else None
    if return_sweeps:   # (..., N) int32: QR sweeps per active block size m (index m - 1)
        sweeps = np.zeros(a.shape[:-1], np.int32)
        _chk(lib().nfm_oracle_qr_eig_sym_sweeps(_dt(a), N, ctypes.c_int64(_nb(a, 2)), int(upper), int(compute_u),
                                                int(max_iter), ctypes.c_double(tol), _p(a), _p(vals), _p(vecs),
                                                _p(sweeps)))
        return ((vals, vecs) if compute_u else vals), sweeps
    _chk(lib().nfm_oracle_qr_eig_sym(_dt(a), N, ctypes.c_int64(_nb(a, 2)), int(upper), int(compute_u), int(max_iter),
                                     ctypes.c_double(tol), _p(a), _p(vals), _p(vecs)))
    return (vals, vecs) if compute_u else vals


def median(x, axis=None, omitnan=False):
    """`median` (`reduce.py:384-428` = torch.median after moving the reduced dims last): the LOWER
    median (rank (count - 1) // 2) and the first position holding it (signed zeros: -0.0 sorts
    before +0.0, they stay equal as numbers); a NaN propagates unless omitnan (the docstring's
    intent, quirk Q14).  Plain numpy sort: small cases only.
    Returns (values, flat indices into the reduced dims)."""
    x = np.asarray(x)
    if axis is None:
        rows = x.reshape(1, -1)
        shape = ()
    else:
        axes = tuple(a % x.ndim for a in (axis if isinstance(axis, (tuple, list)) else (axis,)))
        kept = [d for d in range(x.ndim) if d not in axes]
        rows = np.transpose(x, kept + list(axes)).reshape(int(np.prod([x.shape[d] for d in kept], dtype=np.int64)), -1)
        shape = tuple(x.shape[d] for d in kept)
    vals = np.empty(len(rows), x.dtype)
    idx = np.zeros(len(rows), np.int64)
    for r, row in enumerate(rows):
        nan = np.isnan(row)
        if (nan.any() and not omitnan) or nan.all():
            vals[r] = np.nan
            idx[r] = int(np.argmax(nan)) if nan.any() and not omitnan else 0
            continue
        v = row[~nan]
        v = v[np.lexsort((~np.signbit(v), v))]          # ascending, -0.0 before +0.0 (equal as numbers)
        vals[r] = v[(len(v) - 1) // 2]
        idx[r] = int(np.argmax((row == vals[r]) & (np.signbit(row) == np.signbit(vals[r]))))
    return vals.reshape(shape), idx.reshape(shape)
