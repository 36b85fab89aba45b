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
           'sym_outer', 'sym_matmul', 'batch_inv', 'batch_det', 'batch_matvec', 'reduce',
           'set_num_threads']

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libnfm_oracle.so')
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = [os.path.join(_HERE, f) for f in ('nfm_oracle.c', 'nfm_oracle_body.inc')]
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
