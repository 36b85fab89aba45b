"""
Givens / Householder QR and the symmetric eigensolver for large batches of small matrices
on MI355X -- drop-in for `nitorch_fastmath.qr` (`qr.py:1-11`), same names, argument
order and defaults.

`torch.linalg.eigh` is very slow for large batches of tiny matrices (Hessian filters on
images: one 3x3 per voxel).  The reference implements the explicit QR algorithm as whole-batch
TorchScript ops with a batch-global convergence test (three host syncs per iteration); here
each matrix lives in the registers of one lane and iterates until ITS OWN convergence
(`_impl/qr.py:640-653`'s criterion, applied per matrix), one kernel launch per call.

Real dtypes (float32 / float64) only.  Deliberate deviations from upstream, all bug fixes
(SURVEY quirks Q7-Q9): `eig_sym`/`hessenberg_sym` work for every order <= 16 (upstream
raises for batched n > 5); `rq_hessenberg` returns the true R Q for any Hessenberg input
(upstream is only right for tridiagonal input or n <= 3); eigenvalues come in each matrix's
own deflation order, i.e. what upstream returns when called on that matrix alone (its order
for a batch depends on the other matrices in the batch).

`eig_sym` has two arithmetic modes (`arithmetic=`, default `SWEEP_ARITHMETIC` = `'reference'`):
`'reference'` keeps the reference's operation order, tolerance and correctly rounded division /
square root and reproduces the CPU path bit for bit -- values, deflation ORDER and eigenvector
SIGNS (the division and square root sequences are the IEEE ones without their range scaling,
taken on a wavefront vote that every operand is in range; anything else takes the full sequence);
`'fast'` (opt-in) runs the QR sweeps on v_rsq + Newton steps with fma contraction, diagonalises the
last 2x2 block in closed form (one Jacobi rotation) and never deflates below the working precision
of the dtype (`tol` is floored at (eps/4)^2: float32 stops at |e| <= 1.5e-8 |d| where the
reference's default 1e-32 asks for 1e-16 |d|) -- 2-3x the throughput and the same accuracy
against the exact eigenvalues, but the deflation order and the eigenvector signs differ from the
reference's for a share of the matrices (float32 3x3: 2 %, 8x8: 35 %; the number of sweeps a stage
takes decides where the remaining eigenvalues land) and float32 values by up to 1.4e-6.
"""
__all__ = [
    'eig_sym',
    'qr_hessenberg',
    'rq_hessenberg',
    'hessenberg',
    'hessenberg_sym',
    'householder',
    'householder_apply',
    'givens',
    'givens_apply',
]
import ctypes
import torch
from . import _lib
from ._dispatch import same_dtype, on_device, Batch, dtype_code, expand_batch, no_grad_required, require_gpu, stream_ptr, broadcast_shapes
from .utils import ensure_list

# default arithmetic of the QR sweeps of eig_sym: 'fast' or 'reference' (module docstring)
SWEEP_ARITHMETIC = 'reference'


def _prep(*tensors):
    tensors = [torch.as_tensor(t) for t in tensors]
    dev = require_gpu(*tensors)
    no_grad_required(*tensors)
    dtype = tensors[0].dtype
    for t in tensors[1:]:
        dtype = torch.promote_types(dtype, t.dtype)
    if dtype.is_complex:
        raise TypeError('nitorch_fastmath_amd.qr supports real float32/float64 matrices only')
    dtype_code(dtype)
    return dev, dtype, same_dtype(tensors, dtype)


def _check_finite(check, *tensors):
    if check:
        for t in tensors:
            if t is not None and not torch.isfinite(t).all():
                raise ValueError('Input has non finite values.')


def _check_square(a):
    if a.dim() < 2 or a.shape[-1] != a.shape[-2]:
        raise ValueError('Expected square matrix. Got ({}, {})'.format(a.shape[-2], a.shape[-1]))
    if not 1 <= a.shape[-1] <= _lib.MAX_DIM:
        raise ValueError(f'matrix order {a.shape[-1]} outside the supported range 1..{_lib.MAX_DIM}')


def _packed(batch, rec, dtype, dev):
    return torch.empty(tuple(batch) + (rec,), dtype=dtype, device=dev)


def _dummy(batch, dtype, dev):
    # a zero-stride stand-in for the output slot of Batch (the packed buffer is passed raw)
    return torch.empty((), dtype=dtype, device=dev).expand(tuple(batch))


def _run(fn, args_before, batch, inputs, ncomp, dtype, dev, out):
    """Collapse the batch of `inputs`, call `fn(*args_before, n_outer, n_inner, *operands, out_ptr, stream)`."""
    b = Batch(batch, list(inputs) + [_dummy(batch, dtype, dev)], list(ncomp) + [0])
    ops = [ctypes.byref(o) if o is not None else None for o in b.operands[:-1]]
    with on_device(dev):
        _lib.check(fn(*args_before, b.n_outer, b.n_inner, *ops, out.data_ptr(), stream_ptr(dev)))


def _unpack_reflectors(pack, n):
    """(..., n-2, n-1) zero-padded slots -> list of views (..., n-1-k)."""
    return [pack[..., k, :n - 1 - k] for k in range(max(n - 2, 0))]


def eig_sym(a, compute_u=False, upper=True, inplace=False, check_finite=True, max_iter=1024, tol=1e-32, *,
            arithmetic=None):
    """Compute the eigendecomposition of a symmetric square matrix (`qr.py:30-100`).

    Eigenvalues are **not** sorted (deflation order).

    Parameters
    ----------
    a : `(..., m, m) tensor`
    compute_u : `bool`, default=False
        Compute the eigenvectors. If False, only return ``s``.
    upper : `bool`, default=True
        Whether to use the upper or lower triangular component.
    inplace : `bool`, default=False
        Accepted for compatibility (the input is never modified).
    check_finite : `bool`, default=True
    max_iter : `int`, default=1024
    tol : `float`, default=1e-32
        deflate when e^2 <= tol (d0^2 + d1^2); `arithmetic='fast'` uses max(tol, (eps/4)^2)
    arithmetic : `{'reference', 'fast'}`, keyword-only, default=`SWEEP_ARITHMETIC` (`'reference'`)
        extension, see the module docstring: `'reference'` reproduces the reference CPU path bit
        for bit (deflation order and eigenvector signs included); `'fast'` trades that for speed.

    Returns
    -------
    s : `(..., m) tensor`
    u : `(..., m, m) tensor`, optional
    """
    from ._autograd import EigSymFn, needs_grad
    arithmetic = SWEEP_ARITHMETIC if arithmetic is None else arithmetic
    if arithmetic not in ('fast', 'reference'):
        raise ValueError(f"arithmetic must be 'fast' or 'reference', got {arithmetic!r}")
    if needs_grad(a):
        a = torch.as_tensor(a)
        _check_finite(check_finite, a.detach())
        _check_square(a)
        return EigSymFn.apply(a, bool(compute_u), bool(upper), int(max_iter), float(tol), arithmetic)
    dev, dtype, (a,) = _prep(a)
    _check_finite(check_finite, a)
    _check_square(a)
    n = a.shape[-1]
    batch = a.shape[:-2]
    out = _packed(batch, n + (n * n if compute_u else 0), dtype, dev)
    L = _lib.lib()
    flags = (_lib.EIG_VECTORS if compute_u else 0) | (_lib.EIG_FAST if arithmetic == 'fast' else 0)
    _run(L.nfm_qr_eig_sym, (dtype_code(dtype), n, int(bool(upper)), flags, int(max_iter), float(tol)),
         batch, [a], [2], dtype, dev, out)
    if compute_u:
        return out[..., :n], out[..., n:].unflatten(-1, (n, n))
    return out


def rq_hessenberg(h, u=None, inplace=False, check_finite=True):
    """Compute the QR decomposition of a Hessenberg matrix and ``R @ Q`` (`qr.py:103-142`).

    Returns ``h' = R Q`` and, if ``u`` is given, ``u' = u Q``.
    """
    if u is None:
        dev, dtype, (h,) = _prep(h)
    else:
        dev, dtype, (h, u) = _prep(h, u)
    _check_finite(check_finite, h)
    _check_square(h)
    n = h.shape[-1]
    batch = h.shape[:-2] if u is None else broadcast_shapes(h.shape[:-2], u.shape[:-2])
    out = _packed(batch, n * n * (2 if u is not None else 1), dtype, dev)
    L = _lib.lib()
    if u is None:
        b = Batch(batch, [expand_batch(batch, h, 2), _dummy(batch, dtype, dev)], [2, 0])
        with on_device(dev):
            _lib.check(L.nfm_qr_rq_hessenberg(dtype_code(dtype), n, 0, b.n_outer, b.n_inner,
                                              ctypes.byref(b.operands[0]), None, out.data_ptr(), stream_ptr(dev)))
        return out.unflatten(-1, (n, n))
    b = Batch(batch, [expand_batch(batch, h, 2), expand_batch(batch, u, 2), _dummy(batch, dtype, dev)], [2, 2, 0])
    with on_device(dev):
        _lib.check(L.nfm_qr_rq_hessenberg(dtype_code(dtype), n, 0, b.n_outer, b.n_inner,
                                          ctypes.byref(b.operands[0]), ctypes.byref(b.operands[1]),
                                          out.data_ptr(), stream_ptr(dev)))
    out = out.unflatten(-1, (2, n, n))
    return out[..., 0, :, :], out[..., 1, :, :]


def qr_hessenberg(h, inplace=False, check_finite=True):
    """QR decomposition of a Hessenberg matrix by Givens rotations (`qr.py:145-181`): returns (q, r)."""
    dev, dtype, (h,) = _prep(h)
    _check_finite(check_finite, h)
    _check_square(h)
    n = h.shape[-1]
    batch = h.shape[:-2]
    out = _packed(batch, 2 * n * n, dtype, dev)
    _run(_lib.lib().nfm_qr_qr_hessenberg, (dtype_code(dtype), n), batch, [h], [2], dtype, dev, out)
    out = out.unflatten(-1, (2, n, n))
    return out[..., 0, :, :], out[..., 1, :, :]


def _hessenberg(a, sym, upper, with_u, check_finite):
    dev, dtype, (a,) = _prep(a)
    _check_finite(check_finite, a)
    _check_square(a)
    n = a.shape[-1]
    batch = a.shape[:-2]
    nu = max(n - 2, 0) * (n - 1) if with_u else 0
    out = _packed(batch, n * n + nu, dtype, dev)
    _run(_lib.lib().nfm_qr_hessenberg, (dtype_code(dtype), n, int(sym), int(bool(upper)), int(bool(with_u))),
         batch, [a], [2], dtype, dev, out)
    h = out[..., :n * n].unflatten(-1, (n, n))
    if with_u:
        pack = out[..., n * n:].unflatten(-1, (max(n - 2, 0), n - 1)) if n > 2 else None
        return h, (_unpack_reflectors(pack, n) if pack is not None else [])
    return h, None


def hessenberg(a, inplace=False, check_finite=True, compute_u=False):
    """Hessenberg form of the matrix (or matrices) ``a`` (`qr.py:184-223`).

    Returns ``h`` or ``(h, u)`` with ``u`` the list of Householder reflectors.
    """
    h, u = _hessenberg(a, 0, True, compute_u, check_finite)
    return (h, u) if compute_u else h


def hessenberg_sym(a, upper=True, fill=True, inplace=False, check_finite=True, compute_u=False):
    """Tridiagonal form of the symmetric matrix (or matrices) ``a`` (`qr.py:226-275`).

    Only the ``upper`` (or lower) triangle of ``a`` is read.  With ``fill=False`` the other
    triangle of the result keeps the input's values, as upstream.
    """
    a = torch.as_tensor(a)
    h, u = _hessenberg(a, 1, upper, compute_u, check_finite)
    if not fill:
        n = a.shape[-1]
        keep = torch.ones(n, n, dtype=torch.bool, device=h.device)
        keep = torch.triu(keep) if upper else torch.tril(keep)
        h = torch.where(keep, h, a.to(h.dtype))
    return (h, u) if compute_u else h


def householder(x, basis=0, inplace=False, check_finite=True, return_alpha=False):
    """Householder reflector of a vector (`qr.py:278-327`).

    Returns ``u`` (and ``alpha``, the projection of ``x`` on the Euclidean basis, on request).
    """
    dev, dtype, (x,) = _prep(x)
    _check_finite(check_finite, x)
    n = x.shape[-1]
    if not 1 <= n <= _lib.MAX_DIM:
        raise ValueError(f'vector length {n} outside the supported range 1..{_lib.MAX_DIM}')
    basis = basis if basis >= 0 else n + basis
    batch = x.shape[:-1]
    out = _packed(batch, n + 1, dtype, dev)
    _run(_lib.lib().nfm_qr_householder, (dtype_code(dtype), n, int(basis)), batch, [x], [1], dtype, dev, out)
    u, alpha = out[..., :n], out[..., n]
    return (u, alpha) if return_alpha else u


def householder_apply(a, u, k=None, side='both', inverse=False, inplace=False, check_finite=True):
    """Apply a series of Householder reflectors to a matrix (`qr.py:330-372`).

    Each reflector ``u_k`` of length ``m`` acts on the trailing ``m`` rows / columns.
    """
    a = torch.as_tensor(a)
    us = ensure_list(u)
    dev, dtype, ts = _prep(a, *us)
    a, us = ts[0], ts[1:]
    _check_finite(check_finite, a)
    _check_square(a)
    if side.lower() not in _lib.SIDE:
        raise ValueError(f'unknown side {side}')
    n = a.shape[-1]
    batch = broadcast_shapes(a.shape[:-2], *[uk.shape[:-1] for uk in us])
    out = expand_batch(batch, a, 2).clone(memory_format=torch.contiguous_format)
    if inverse:
        us = us[::-1]
    L = _lib.lib()
    for uk in us:
        m = uk.shape[-1]
        b = Batch(batch, [expand_batch(batch, uk, 1), out], [1, 2])
        with on_device(dev):
            _lib.check(L.nfm_qr_householder_apply(dtype_code(dtype), n, m, _lib.SIDE[side.lower()], b.n_outer, b.n_inner,
                                                  ctypes.byref(b.operands[1]), ctypes.byref(b.operands[0]),
                                                  stream_ptr(dev)))
    if inplace and out.shape == a.shape:
        a.copy_(out)
        return a
    return out


def givens(x, y):
    r"""Givens rotation: ``c = x / norm([x, y])``, ``s = -y / norm([x, y])`` (`qr.py`, `_impl/qr.py:326-369`)."""
    dev, dtype, (x, y) = _prep(x, y)
    batch = broadcast_shapes(x.shape, y.shape)
    out = _packed(batch, 2, dtype, dev)
    _run(_lib.lib().nfm_qr_givens, (dtype_code(dtype),), batch,
         [expand_batch(batch, x, 0), expand_batch(batch, y, 0)], [0, 0], dtype, dev, out)
    return out[..., 0], out[..., 1]


def givens_apply(a, c, s, i=0, j=None, side='both', inplace=False, check_finite=True):
    """Apply a Givens rotation to rows and/or columns ``i`` and ``j`` of a matrix (`qr.py:375-424`).

    ``c`` and ``s`` must broadcast against ``a[..., i, :]`` (e.g. shape ``(..., 1)``), as upstream.
    """
    dev, dtype, (a, c, s) = _prep(a, c, s)
    _check_finite(check_finite, a)
    _check_square(a)
    if side.lower() not in _lib.SIDE:
        raise ValueError(f'unknown side {side}')
    n = a.shape[-1]
    j = i + 1 if j is None else j
    i = i if i >= 0 else n + i
    j = j if j >= 0 else n + j
    vshape = broadcast_shapes(a.shape[:-2] + (n,), c.shape, s.shape)
    batch = vshape[:-1]
    out = expand_batch(batch, a, 2).clone(memory_format=torch.contiguous_format)
    b = Batch(batch, [c.expand(vshape), s.expand(vshape), out], [1, 1, 2])
    with on_device(dev):
        _lib.check(_lib.lib().nfm_qr_givens_apply(dtype_code(dtype), n, _lib.SIDE[side.lower()], int(i), int(j),
                                                  b.n_outer, b.n_inner, ctypes.byref(b.operands[2]),
                                                  ctypes.byref(b.operands[0]), ctypes.byref(b.operands[1]),
                                                  stream_ptr(dev)))
    if inplace and out.shape == a.shape:
        a.copy_(out)
        return a
    return out
