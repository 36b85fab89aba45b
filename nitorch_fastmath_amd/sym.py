"""
Compact-storage symmetric matrices on MI355X -- drop-in for `nitorch_fastmath.sym`.

Same layout as the reference (`nitorch_fastmath/sym.py:7-14`): the flattened matrix
holds the diagonal first, then the rows of the upper triangle::

    [ a d e ]
    [ . b f ]   =>  [a b c d e f]
    [ . . c ]

Matrix-vector functions (`sym_matvec`, `sym_solve`, and the add/sub forms) also accept
and auto-detect compact diagonal / scaled-identity / full matrices, as the reference
documents (`sym.py:16-24`): for a vector `(*, N)` and a matrix `(*, NN)`, `NN` may be
`1` (scaled identity), `N` (diagonal), `N*(N+1)//2` (symmetric) or `N*N` (full).

Every function launches hand-written gfx950 kernels through the C ABI of
`libnfm_hip.so` on the current stream; tensors must live on the GPU.
"""
__all__ = [
    'sym_to_full', 'sym_diag', 'sym_outer', 'sym_det', 'sym_matmul',
    'sym_matvec',
    'sym_addmatvec', 'sym_addmatvec_',
    'sym_submatvec', 'sym_submatvec_',
    'sym_solve', 'sym_solve_',
    'sym_invert', 'sym_invert_',
    'sym_matmul_solve',      # extension: fused Gauss-Newton step (not in the reference)
]
import ctypes
from math import sqrt
import torch
from . import _lib
from ._dispatch import (same_dtype, on_device, Batch, broadcast_shapes, common_dtype, dtype_code, expand_batch, no_grad_required,
                        require_gpu, stream_ptr)


def _nb_prm(K):
    M = int((sqrt(1 + 8 * K) - 1) // 2)
    if M * (M + 1) // 2 != K:
        raise ValueError(f'last dimension {K} is not M*(M+1)/2 for any integer M')
    return M


def _check_order(M):
    if not 1 <= M <= _lib.MAX_DIM:
        raise ValueError(f'matrix order {M} outside the supported range 1..{_lib.MAX_DIM}')


def _mat_kind(NN, N):
    """`sym.py:16-24`; ambiguous sizes (N = 1) resolve to the symmetric reading."""
    if NN == N * (N + 1) // 2:
        return _lib.MAT_SYM
    if NN == N:
        return _lib.MAT_DIAG
    if NN == 1:
        return _lib.MAT_SCAL
    if NN == N * N:
        return _lib.MAT_FULL
    raise ValueError(f'matrix with {NN} components does not match a vector of length {N}: '
                     f'expected 1, {N}, {N * (N + 1) // 2} or {N * N}')


# orders 9..16 of sym_solve / sym_invert: 'auto' = positive definite first, 'always' = pivoted at once
PIVOTING = 'auto'


def _pivoting(pivoting):
    mode = PIVOTING if pivoting is None else pivoting
    if mode not in ('auto', 'always'):
        raise ValueError(f"pivoting must be 'auto' or 'always', got {mode!r}")
    return mode == 'always'


def _prep(dtype, *tensors):
    tensors = [None if t is None else torch.as_tensor(t) for t in tensors]
    dev = require_gpu(*tensors)
    no_grad_required(*tensors)
    dtype = common_dtype(dtype, *tensors)
    dtype_code(dtype)
    return dev, dtype, same_dtype(tensors, dtype)


def _alloc_out(out, shape, dtype, device, like=None):
    if out is None:
        # a channel-first (component-major) operand of the output's shape hands its layout on:
        # the whole call then runs through the SoA tiles and downstream ops keep the field layout
        if (like is not None and tuple(like.shape) == tuple(shape) and like.dim() >= 2
                and like.stride(-1) != 1 and not like.is_contiguous()
                and like.numel() > 0 and 0 not in like.stride()):
            cand = torch.empty_like(like, dtype=dtype)      # preserve_format keeps dense strides
            if cand.stride() == like.stride():
                return cand, None
        return torch.empty(shape, dtype=dtype, device=device), None
    if tuple(out.shape) != tuple(shape):
        raise ValueError(f'out has shape {tuple(out.shape)}, expected {tuple(shape)}')
    if out.dtype != dtype or out.device != device:
        raise ValueError('out must have the computation dtype and live on the same device')
    return out, None


def _full_view(mat, N, kind):
    """Full matrices come as (..., N*N) per `sym.py:24`; view them as (..., N, N)."""
    if kind == _lib.MAT_FULL:
        return mat.unflatten(-1, (N, N)), 2
    return mat, 1


def _matvec_impl(mode, inp, mat, vec, dtype, out):
    dev, dtype, (inp, mat, vec) = _prep(dtype, inp, mat, vec)
    N = vec.shape[-1]
    kind = _mat_kind(mat.shape[-1], N)
    if N > _lib.MAX_DIM:          # the reference's own large-order route, on the device (_bigorder.py)
        from . import _bigorder
        return _bigorder.sym_matvec(mode, inp, mat, vec, out, mat.shape[-1])
    matv, mat_nc = _full_view(mat, N, kind)
    shapes = [mat.shape[:-1], vec.shape[:-1]] + ([inp.shape[:-1]] if inp is not None else [])
    batch = broadcast_shapes(*shapes)
    if inp is not None and inp.shape[-1] != N:
        raise ValueError('inp and vec must have the same number of components')
    out, _ = _alloc_out(out, tuple(batch) + (N,), dtype, dev, like=vec if (N <= 8 or kind == _lib.MAT_SYM) else None)
    ops = [expand_batch(batch, matv, mat_nc), expand_batch(batch, vec, 1)]
    ncs = [mat_nc, 1]
    if inp is not None:
        ops.append(expand_batch(batch, inp, 1))
        ncs.append(1)
    ops.append(out)
    ncs.append(1)
    b = Batch(batch, ops, ncs, pack=N > 8 and kind == _lib.MAT_SYM)
    o = b.operands
    o_inp = ctypes.byref(o[2]) if inp is not None else None
    with on_device(dev):
        _lib.check(_lib.lib().nfm_sym_matvec(
            dtype_code(dtype), N, kind, mode, b.n_outer, b.n_inner, ctypes.byref(o[0]),
            ctypes.byref(o[1]), o_inp, ctypes.byref(o[-1]), stream_ptr(dev)))
    b.finish()
    return out


def _cast(dtype, *tensors):
    """differentiable torch route of the orders > MAX_DIM: operands as tensors of the computation dtype"""
    ts = [None if t is None else torch.as_tensor(t) for t in tensors]
    if dtype is None:
        dtype = ts[0].dtype
        for t in ts[1:]:
            if t is not None:
                dtype = torch.promote_types(dtype, t.dtype)
    return [None if t is None else t.to(dtype) for t in ts]


def _matvec(mode, inp, mat, vec, dtype, out):
    from ._autograd import SymMatvecFn, needs_grad
    if needs_grad(inp, mat, vec):
        if out is not None:
            raise RuntimeError('out= is not supported for tensors that require grad')
        if torch.as_tensor(vec).shape[-1] > _lib.MAX_DIM:   # torch.linalg route: torch differentiates it (forward AND backward)
            from . import _bigorder
            mat_, vec_, inp_ = _cast(dtype, mat, vec, inp)
            return _bigorder.sym_matvec(mode, inp_, mat_, vec_, None, mat_.shape[-1])
        return SymMatvecFn.apply(mode, inp, torch.as_tensor(mat), torch.as_tensor(vec), dtype)
    return _matvec_impl(mode, inp, mat, vec, dtype, out)


def sym_matvec(mat, vec, dtype=None, out=None):
    r"""Matrix-vector product with a compact symmetric matrix: `mat @ vec`.

    Replaces `nitorch_fastmath.sym.sym_matvec` (in-repo: `_impl/sym.py:134-172`).

    Parameters
    ----------
    mat : `(..., NN) tensor`
        Compact matrix; `NN` in `{1, M, M*(M+1)//2, M*M}` (see module docstring).
    vec : `(..., M) tensor`
    dtype : `torch.dtype`, optional
        Computation (and output) dtype; default: promoted input dtype.
    out : `(..., M) tensor`, optional

    Returns
    -------
    matvec : `(..., M) tensor`
    """
    return _matvec(0, None, mat, vec, dtype, out)


def sym_addmatvec(inp, mat, vec, dtype=None, out=None):
    """`inp + mat @ vec` (reference name list `sym.py:31`)."""
    return _matvec(+1, inp, mat, vec, dtype, out)


def sym_addmatvec_(inp, mat, vec):
    """In-place `inp += mat @ vec`."""
    _require_inplace_ok(inp, broadcast_shapes(mat.shape[:-1], vec.shape[:-1]) + vec.shape[-1:])
    return _matvec_impl(+1, inp, mat, vec, inp.dtype, inp)


def sym_submatvec(inp, mat, vec, dtype=None, out=None):
    """`inp - mat @ vec` (reference name list `sym.py:32`)."""
    return _matvec(-1, inp, mat, vec, dtype, out)


def sym_submatvec_(inp, mat, vec):
    """In-place `inp -= mat @ vec`."""
    _require_inplace_ok(inp, broadcast_shapes(mat.shape[:-1], vec.shape[:-1]) + vec.shape[-1:])
    return _matvec_impl(-1, inp, mat, vec, inp.dtype, inp)


def _require_inplace_ok(t, shape):
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f'in-place operand has shape {tuple(t.shape)} but the result has shape '
                         f'{tuple(shape)} (it cannot be broadcast)')


def sym_solve(mat, vec, eps=None, dtype=None, out=None, *, pivoting=None):
    r"""Left matrix division for compact symmetric matrices: `mat \ vec`.

    Replaces `nitorch_fastmath.sym.sym_solve` (in-repo: `_impl/sym.py:327-398`).
    Orders up to 4 use the reference's closed forms evaluated in its operation order
    (bit-identical to its CPU path); larger orders use LU with partial pivoting per
    lane, like the `torch.linalg.solve` branch of the reference.

    Parameters
    ----------
    mat : `(..., NN) tensor`
    vec : `(..., M) tensor`
    eps : `float or (M,) sequence[float]`, optional
        Smoothing term added to the diagonal of `mat` (last value repeated).
    dtype, out : optional
    pivoting : {'auto', 'always'}, keyword-only, default `sym.PIVOTING` = 'auto'
        Orders 9..16.  'auto': the unpivoted LDL^T first -- positive definite matrices, what Hessians are -- and the
        pivoted elimination for the groups of matrices that are not; 'always': the pivoted elimination at once (a
        batch of INDEFINITE matrices pays the attempt for nothing: up to 2x).  Same answers within rounding.

    Returns
    -------
    result : `(..., M) tensor`
    """
    from ._autograd import SymSolveFn, needs_grad
    if needs_grad(mat, vec):
        if out is not None:
            raise RuntimeError('out= is not supported for tensors that require grad')
        if torch.as_tensor(vec).shape[-1] > _lib.MAX_DIM:   # torch.linalg route: torch differentiates it
            from . import _bigorder
            mat_, vec_ = _cast(dtype, mat, vec)
            return _bigorder.sym_solve(mat_, vec_, eps, None, mat_.shape[-1])
        return SymSolveFn.apply(torch.as_tensor(mat), torch.as_tensor(vec), eps, dtype)
    dev, dtype, (mat, vec) = _prep(dtype, mat, vec)
    N = vec.shape[-1]
    kind = _mat_kind(mat.shape[-1], N)
    if N > _lib.MAX_DIM:          # densify + torch.linalg.solve on the device, as `_impl/sym.py:392-396`
        from . import _bigorder
        return _bigorder.sym_solve(mat, vec, eps, out, mat.shape[-1])
    matv, mat_nc = _full_view(mat, N, kind)
    batch = broadcast_shapes(mat.shape[:-1], vec.shape[:-1])
    # (orders 9..16 of compact matrices take component-major fields as they are since round 3 -- `spd_strided_kernel`
    # -- so channel-first input gets channel-first output at every order; Batch packs what is strided along the batch)
    piv = _pivoting(pivoting)
    out, _ = _alloc_out(out, tuple(batch) + (N,), dtype, dev,
                        like=vec if (N <= 8 or (kind == _lib.MAT_SYM and not piv)) else None)
    b = Batch(batch, [expand_batch(batch, matv, mat_nc), expand_batch(batch, vec, 1), out], [mat_nc, 1, 1],
              pack=('all' if piv else True) if (N > 8 and kind == _lib.MAT_SYM) else False)
    o = b.operands
    eps_p = None
    if eps is not None:
        e = [float(x) for x in torch.as_tensor(eps, dtype=torch.float64).flatten().tolist()]
        if not e:
            raise ValueError('eps is empty')
        e = (e + [e[-1]] * N)[:N]
        eps_p = (ctypes.c_double * _lib.MAX_DIM)(*(e + [0.0] * (_lib.MAX_DIM - N)))
    with on_device(dev):
        _lib.check(_lib.lib().nfm_sym_solve(
            dtype_code(dtype), N, kind | (_lib.MAT_PIVOTED if piv else 0), b.n_outer, b.n_inner, ctypes.byref(o[0]),
            ctypes.byref(o[1]), ctypes.byref(o[2]), eps_p, stream_ptr(dev)))
    b.finish()
    return out


def sym_solve_(mat, vec, eps=None):
    """In-place `sym_solve`: overwrites `vec` with `mat \\ vec` (`sym.py:33`)."""
    _require_inplace_ok(vec, broadcast_shapes(mat.shape[:-1], vec.shape[:-1]) + vec.shape[-1:])
    return sym_solve(mat, vec, eps=eps, dtype=vec.dtype, out=vec)


def sym_invert(mat, diag=False, dtype=None, out=None, *, pivoting=None):
    r"""Inverse of compact symmetric matrices, returned in compact storage.

    Replaces `nitorch_fastmath.sym.sym_invert` (in-repo: `_impl/sym.py:455-493`,
    which runs M full solves; here one factorisation per matrix).

    Parameters
    ----------
    mat : `(..., M*(M+1)//2) tensor`
    diag : `bool`, default=False
        If True, only return the diagonal of the inverse, shape `(..., M)`.
    pivoting : {'auto', 'always'}, keyword-only: see `sym_solve`.
    """
    from ._autograd import SymInvertFn, needs_grad
    if needs_grad(mat):
        if out is not None:
            raise RuntimeError('out= is not supported for tensors that require grad')
        if _nb_prm(torch.as_tensor(mat).shape[-1]) > _lib.MAX_DIM:
            from . import _bigorder
            (mat_,) = _cast(dtype, mat)
            return _bigorder.sym_invert(mat_, _nb_prm(mat_.shape[-1]), bool(diag), None)
        return SymInvertFn.apply(torch.as_tensor(mat), bool(diag), dtype)
    dev, dtype, (mat,) = _prep(dtype, mat)
    M = _nb_prm(mat.shape[-1])
    if M > _lib.MAX_DIM:
        from . import _bigorder
        return _bigorder.sym_invert(mat, M, bool(diag), out)
    batch = mat.shape[:-1]
    piv = _pivoting(pivoting)
    uncovered = piv          # (the pivoted kernels of orders 9..16 want contiguous records: everything is packed for them)
    out, _ = _alloc_out(out, tuple(batch) + ((M,) if diag else (mat.shape[-1],)), dtype, dev, like=None if uncovered else mat)
    b = Batch(batch, [mat, out], [1, 1], pack=('all' if uncovered else True) if (M > 8 and not diag) else False)
    o = b.operands
    with on_device(dev):
        _lib.check(_lib.lib().nfm_sym_invert(
            dtype_code(dtype), M, int(bool(diag)) | (_lib.INVERT_PIVOTED if piv else 0), b.n_outer, b.n_inner, ctypes.byref(o[0]),
            ctypes.byref(o[1]), stream_ptr(dev)))
    b.finish()
    return out


def sym_invert_(mat):
    """In-place `sym_invert`: overwrites `mat` with its compact inverse (`sym.py:34`)."""
    return sym_invert(mat, dtype=mat.dtype, out=mat)


def sym_det(mat, dtype=None, out=None):
    r"""Determinant of compact symmetric matrices (`_impl/sym.py:401-452`).

    The reference derives M from a batch dimension by mistake (quirk Q2); this
    implementation uses the compact dimension, as documented.
    """
    from ._autograd import SymDetFn, needs_grad
    if needs_grad(mat):
        if out is not None:
            raise RuntimeError('out= is not supported for tensors that require grad')
        if _nb_prm(torch.as_tensor(mat).shape[-1]) > _lib.MAX_DIM:
            from . import _bigorder
            (mat_,) = _cast(dtype, mat)
            return _bigorder.sym_det(mat_, _nb_prm(mat_.shape[-1]), None)
        return SymDetFn.apply(torch.as_tensor(mat), dtype)
    dev, dtype, (mat,) = _prep(dtype, mat)
    M = _nb_prm(mat.shape[-1])
    if M > _lib.MAX_DIM:
        from . import _bigorder
        return _bigorder.sym_det(mat, M, out)
    batch = mat.shape[:-1]
    out, _ = _alloc_out(out, tuple(batch), dtype, dev)
    b = Batch(batch, [mat, out], [1, 0], pack=M > 8)
    o = b.operands
    with on_device(dev):
        _lib.check(_lib.lib().nfm_sym_det(
            dtype_code(dtype), M, b.n_outer, b.n_inner, ctypes.byref(o[0]), ctypes.byref(o[1]),
            stream_ptr(dev)))
    b.finish()
    return out


def _grad_guard(out):
    if out is not None:
        raise RuntimeError('out= is not supported for tensors that require grad')


def sym_to_full(mat, dtype=None, out=None):
    r"""Compact symmetric `(..., M*(M+1)//2)` -> full `(..., M, M)` (`_impl/sym.py:16-60`)."""
    from ._autograd import SymToFullFn, needs_grad
    if needs_grad(mat):
        _grad_guard(out)
        return SymToFullFn.apply(torch.as_tensor(mat), dtype)
    dev, dtype, (mat,) = _prep(dtype, mat)
    M = _nb_prm(mat.shape[-1])
    if M > _lib.MAX_DIM:
        from . import _bigorder
        return _bigorder._deliver(_bigorder.to_full(mat, M), out)
    batch = mat.shape[:-1]
    out, _ = _alloc_out(out, tuple(batch) + (M, M), dtype, dev)
    b = Batch(batch, [mat, out], [1, 2])
    o = b.operands
    with on_device(dev):
        _lib.check(_lib.lib().nfm_sym_to_full(
            dtype_code(dtype), M, b.n_outer, b.n_inner, ctypes.byref(o[0]), ctypes.byref(o[1]),
            stream_ptr(dev)))
    b.finish()
    return out


def sym_diag(mat):
    r"""View into the main diagonal of a compact symmetric matrix, shape `(..., M)`
    (`_impl/sym.py:63-84`; a pure view, no kernel)."""
    mat = torch.as_tensor(mat)
    return mat[..., :_nb_prm(mat.shape[-1])]


def sym_outer(x, dtype=None, out=None):
    r"""Symmetric outer product `x x^T` in compact storage (`_impl/sym.py:496-528`)."""
    from ._autograd import SymOuterFn, needs_grad
    if needs_grad(x):
        _grad_guard(out)
        return SymOuterFn.apply(torch.as_tensor(x), dtype)
    dev, dtype, (x,) = _prep(dtype, x)
    M = x.shape[-1]
    _check_order(M)
    batch = x.shape[:-1]
    out, _ = _alloc_out(out, tuple(batch) + (M * (M + 1) // 2,), dtype, dev)
    b = Batch(batch, [x, out], [1, 1])
    o = b.operands
    with on_device(dev):
        _lib.check(_lib.lib().nfm_sym_outer(
            dtype_code(dtype), M, b.n_outer, b.n_inner, ctypes.byref(o[0]), ctypes.byref(o[1]),
            stream_ptr(dev)))
    b.finish()
    return out


def sym_matmul(j, h, dtype=None, out=None):
    r"""Symmetric product `J^T H J` with compact `H`, returned compact (`_impl/sym.py:637-670`).

    j : `(..., k, d)`, h : `(..., k*(k+1)//2)` (or `(..., k)` diagonal) -> `(..., d*(d+1)//2)`.
    For `k == d` in `{2, 3}` the reference's specialised kernels evaluate `J H J^T`
    (quirk Q16); this function returns what the reference returns.
    """
    from ._autograd import SymMatmulFn, needs_grad
    if needs_grad(j, h):
        _grad_guard(out)
        return SymMatmulFn.apply(torch.as_tensor(j), torch.as_tensor(h), dtype)
    dev, dtype, (j, h) = _prep(dtype, j, h)
    k, d = j.shape[-2:]
    _check_order(k)
    _check_order(d)
    if h.shape[-1] == k * (k + 1) // 2:
        hk = _lib.MAT_SYM
    elif h.shape[-1] == k:
        hk = _lib.MAT_DIAG
    else:
        raise ValueError(f'hessian with {h.shape[-1]} components does not match k={k}')
    batch = broadcast_shapes(j.shape[:-2], h.shape[:-1])
    out, _ = _alloc_out(out, tuple(batch) + (d * (d + 1) // 2,), dtype, dev)
    b = Batch(batch, [expand_batch(batch, j, 2), expand_batch(batch, h, 1), out], [2, 1, 1])
    o = b.operands
    with on_device(dev):
        _lib.check(_lib.lib().nfm_sym_matmul(
            dtype_code(dtype), k, d, hk, b.n_outer, b.n_inner, ctypes.byref(o[0]), ctypes.byref(o[1]),
            ctypes.byref(o[2]), stream_ptr(dev)))
    b.finish()
    return out


def sym_matmul_solve(j, h, g, eps=None, dtype=None, out=None):
    r"""EXTENSION (not in the reference): `sym_solve(sym_matmul(j, h), g, eps)` in ONE kernel.

    The Gauss-Newton step of a Hessian field pushed through a Jacobian field,
    `x = (J^T H J + diag(eps))^{-1} g`, without writing the compact `(d x d)` product to HBM and
    reading it back: for k = d = 3 fp32, 84 B per element instead of 132 B.
    Same arithmetic as the two calls, so the result is bit-identical to the chain (including the
    reference's `J H J^T` quirk for k = d in {2, 3}).  Sizes beyond 4 fall back to the chain.

    j : `(..., k, d)`, h : `(..., k*(k+1)//2)` or `(..., k)`, g : `(..., d)` -> `(..., d)`.
    """
    from ._autograd import needs_grad
    k, d = torch.as_tensor(j).shape[-2:]
    if needs_grad(j, h, g) or k > 4 or d > 4:
        return sym_solve(sym_matmul(j, h, dtype=dtype), g, eps=eps, dtype=dtype, out=out)
    dev, dtype, (j, h, g) = _prep(dtype, j, h, g)
    if h.shape[-1] == k * (k + 1) // 2:
        hk = _lib.MAT_SYM
    elif h.shape[-1] == k:
        hk = _lib.MAT_DIAG
    else:
        raise ValueError(f'hessian with {h.shape[-1]} components does not match k={k}')
    if g.shape[-1] != d:
        raise ValueError(f'gradient with {g.shape[-1]} components does not match d={d}')
    batch = broadcast_shapes(j.shape[:-2], h.shape[:-1], g.shape[:-1])
    out, _ = _alloc_out(out, tuple(batch) + (d,), dtype, dev, like=g)
    b = Batch(batch, [expand_batch(batch, j, 2), expand_batch(batch, h, 1), expand_batch(batch, g, 1), out],
              [2, 1, 1, 1])
    o = b.operands
    eps_p = None
    if eps is not None:
        e = [float(x) for x in torch.as_tensor(eps, dtype=torch.float64).flatten().tolist()]
        if not e:
            raise ValueError('eps is empty')
        e = (e + [e[-1]] * d)[:d]
        eps_p = (ctypes.c_double * _lib.MAX_DIM)(*(e + [0.0] * (_lib.MAX_DIM - d)))
    with on_device(dev):
        _lib.check(_lib.lib().nfm_sym_matmul_solve(
            dtype_code(dtype), k, d, hk, b.n_outer, b.n_inner, ctypes.byref(o[0]), ctypes.byref(o[1]),
            ctypes.byref(o[2]), ctypes.byref(o[3]), eps_p, stream_ptr(dev)))
    b.finish()
    return out
