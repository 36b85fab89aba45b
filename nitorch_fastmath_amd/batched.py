"""
Batched determinant / inverse / matrix-vector product for large batches of small
general matrices on MI355X -- drop-in for `nitorch_fastmath.batched`
(`batched.py:16-17`, `_impl/batched.py`).

The reference has closed forms for 1x1..3x3 on the GPU and falls back to torch
(LAPACK / MAGMA-style batched LU) otherwise; here every order 1..16 is one
lane-per-matrix HIP kernel: adjugate closed forms up to 3x3, in-register
Gauss-Jordan / LU with partial pivoting up to 8x8, LDS-resident LU up to 16x16.
"""
__all__ = ['batchmatvec', 'batchdet', 'batchinv']
import ctypes
import torch
from . import _lib
from ._dispatch import (same_dtype, on_device, Batch, broadcast_shapes, common_dtype, dtype_code, expand_batch, no_grad_required,
                        require_gpu, stream_ptr)


def _prep(*tensors):
    tensors = [torch.as_tensor(t) for t in tensors]
    dev = require_gpu(*tensors)
    no_grad_required(*tensors)
    dtype = common_dtype(None, *tensors)
    dtype_code(dtype)
    return dev, dtype, same_dtype(tensors, dtype)


def _like_or_contiguous(like, shape, dtype, dev):
    """Matrix-first / channel-first operands of the output's shape hand their layout on (the call
    then runs through the SoA tiles end to end); everything else gets a contiguous output."""
    if (like is not None and tuple(like.shape) == tuple(shape) and not like.is_contiguous()
            and like.stride(-1) != 1 and like.numel() > 0 and 0 not in like.stride()):
        cand = torch.empty_like(like, dtype=dtype)
        if cand.stride() == like.stride():
            return cand
    return torch.empty(shape, dtype=dtype, device=dev)


def batchdet(a):
    """Batched determinant for large batches of small matrices.

    a : `(..., n, n) tensor` -> `(...) tensor`.  Replaces `_impl/batched.py:35-63`.
    """
    from ._autograd import BatchDetFn, needs_grad
    if needs_grad(a):
        return BatchDetFn.apply(torch.as_tensor(a))
    dev, dtype, (a,) = _prep(a)
    n = a.shape[-1]
    assert a.shape[-2] == n, 'Expected square matrices'
    if n > _lib.MAX_DIM:          # the reference's own route for every order (`_impl/batched.py:53-54`), on the device
        return torch.linalg.det(a)
    batch = a.shape[:-2]
    out = torch.empty(batch, dtype=dtype, device=dev)
    b = Batch(batch, [a, out], [2, 0], pack=n > 8)
    o = b.operands
    with on_device(dev):
        _lib.check(_lib.lib().nfm_batch_det(dtype_code(dtype), n, b.n_outer, b.n_inner,
                                            ctypes.byref(o[0]), ctypes.byref(o[1]), stream_ptr(dev)))
    b.finish()
    return out


def batchinv(a, perturb=False):
    """Batched inversion for large batches of small matrices.

    a : `(..., n, n) tensor` -> `(..., n, n) tensor`.  Replaces `_impl/batched.py:101-130`.

    perturb : bool, default=False
        Reproduce the TorchScript closed forms `inv2`/`inv3` exactly, including their
        determinant perturbation `(max|a| - min|a|) * 1e-12` (`_impl/batched.py:74-76`).
        The default matches the reference's CPU path (`a.inverse()`).
    """
    from ._autograd import BatchInvFn, needs_grad
    if needs_grad(a):
        return BatchInvFn.apply(torch.as_tensor(a), bool(perturb))
    dev, dtype, (a,) = _prep(a)
    n = a.shape[-1]
    assert a.shape[-2] == n, 'Expected square matrices'
    if n > _lib.MAX_DIM:          # `a.inverse()` (`_impl/batched.py:119-120`), on the device
        return torch.linalg.inv(a)
    batch = a.shape[:-2]
    out = _like_or_contiguous(a if n <= 8 else None, tuple(batch) + (n, n), dtype, dev)
    b = Batch(batch, [a, out], [2, 2], pack=n > 8)
    o = b.operands
    flags = _lib.FLAG_TS_PERTURB if perturb else 0
    with on_device(dev):
        _lib.check(_lib.lib().nfm_batch_inv(dtype_code(dtype), n, flags, b.n_outer, b.n_inner,
                                            ctypes.byref(o[0]), ctypes.byref(o[1]), stream_ptr(dev)))
    b.finish()
    return out


def batchmatvec(mat, vec):
    """Batched matrix-vector product for large batches of small matrices.

    mat : `(..., m, n)`, vec : `(..., n)` -> `(..., m)`.  Replaces `_impl/batched.py:154-190`.
    """
    from ._autograd import BatchMatvecFn, needs_grad
    if needs_grad(mat, vec):
        return BatchMatvecFn.apply(torch.as_tensor(mat), torch.as_tensor(vec))
    dev, dtype, (mat, vec) = _prep(mat, vec)
    m, n = mat.shape[-2:]
    if vec.shape[-1] != n:
        raise ValueError(f'matrix {tuple(mat.shape[-2:])} and vector ({vec.shape[-1]},) do not match')
    batch = broadcast_shapes(mat.shape[:-2], vec.shape[:-1])
    out = _like_or_contiguous(vec if m == n else None, tuple(batch) + (m,), dtype, dev)
    b = Batch(batch, [expand_batch(batch, mat, 2), expand_batch(batch, vec, 1), out], [2, 1, 1])
    o = b.operands
    with on_device(dev):
        _lib.check(_lib.lib().nfm_batch_matvec(dtype_code(dtype), m, n, b.n_outer, b.n_inner,
                                               ctypes.byref(o[0]), ctypes.byref(o[1]),
                                               ctypes.byref(o[2]), stream_ptr(dev)))
    b.finish()
    return out
