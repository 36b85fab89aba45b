"""Autograd for the facade: `torch.autograd.Function`s whose forward AND backward run the
HIP kernels of libnfm_hip.so (the backward of a solve is a solve, of a mat-vec a mat-vec,
plus one `nfm_sym_outer2` launch for the compact-matrix cotangent).  Element-wise glue in
the backward of the reductions (mask, broadcast, divide) is plain torch on the device.

Covered: sym_matvec / sym_addmatvec / sym_submatvec, sym_solve, eig_sym (eigenvalues, and
eigenvectors through Giles' formula, as upstream's `_EigSym` `_impl/qr.py:684-735` intends),
sum / nansum / mean / nanmean.  Everything else is forward-only and says so.
"""
import ctypes
import torch
from torch.autograd.function import once_differentiable
from . import _lib
from ._dispatch import Batch, dtype_code, expand_batch, stream_ptr, broadcast_shapes


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def _sum_to(g, shape):
    """Reduce a broadcast gradient back to the operand's shape."""
    if g is None or tuple(g.shape) == tuple(shape):
        return g
    return g.sum_to_size(tuple(shape))


def sym_outer2(x, y, neg=False):
    """compact pull-back of x y^T: out_ii = x_i y_i, out_ij = x_i y_j + x_j y_i (nfm_sym_outer2)."""
    dev, dtype = x.device, x.dtype
    M = x.shape[-1]
    batch = broadcast_shapes(x.shape[:-1], y.shape[:-1])
    out = torch.empty(tuple(batch) + (M * (M + 1) // 2,), dtype=dtype, device=dev)
    b = Batch(batch, [expand_batch(batch, x, 1), expand_batch(batch, y, 1), out], [1, 1, 1])
    o = b.operands
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().nfm_sym_outer2(dtype_code(dtype), M, int(neg), b.n_outer, b.n_inner,
                                             ctypes.byref(o[0]), ctypes.byref(o[1]), ctypes.byref(o[2]),
                                             stream_ptr(dev)))
    return out


def _mat_cotangent(kind, u, v, neg):
    """Cotangent of the (compact / diagonal / scaled-identity / full) matrix operand for a full-matrix
    cotangent u v^T."""
    if kind == _lib.MAT_SYM:
        return sym_outer2(u, v, neg)
    if kind == _lib.MAT_DIAG:
        g = u * v
    elif kind == _lib.MAT_SCAL:
        g = (u * v).sum(-1, keepdim=True)
    else:
        g = (u.unsqueeze(-1) * v.unsqueeze(-2)).flatten(-2)
    return -g if neg else g


def _transposed(mat, kind, N):
    if kind == _lib.MAT_FULL:
        return mat.unflatten(-1, (N, N)).transpose(-1, -2).flatten(-2)
    return mat


class SymMatvecFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mode, inp, mat, vec, dtype):
        from . import sym
        out = sym._matvec_impl(mode, inp, mat, vec, dtype, None)
        ctx.mode = mode
        ctx.kind = sym._mat_kind(mat.shape[-1], vec.shape[-1])
        ctx.shapes = (None if inp is None else inp.shape, mat.shape, vec.shape)
        ctx.dtypes = (None if inp is None else inp.dtype, mat.dtype, vec.dtype)
        ctx.save_for_backward(mat, vec)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import sym
        mat, vec = ctx.saved_tensors
        N = vec.shape[-1]
        sgn = -1 if ctx.mode < 0 else 1
        g = g.contiguous()
        matc, vecc = mat.to(g.dtype), vec.to(g.dtype)
        g_inp = g_mat = g_vec = None
        if ctx.needs_input_grad[1] and ctx.shapes[0] is not None:
            g_inp = _sum_to(g, ctx.shapes[0]).to(ctx.dtypes[0])
        if ctx.needs_input_grad[2]:
            g_mat = _sum_to(_mat_cotangent(ctx.kind, g, vecc, sgn < 0), ctx.shapes[1]).to(ctx.dtypes[1])
        if ctx.needs_input_grad[3]:
            gv = sym._matvec_impl(0, None, _transposed(matc, ctx.kind, N), g, None, None)
            g_vec = _sum_to(gv if sgn > 0 else -gv, ctx.shapes[2]).to(ctx.dtypes[2])
        return None, g_inp, g_mat, g_vec, None


class SymSolveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mat, vec, eps, dtype):
        from . import sym
        x = sym.sym_solve(mat, vec, eps=eps, dtype=dtype)
        ctx.kind = sym._mat_kind(mat.shape[-1], vec.shape[-1])
        ctx.eps = eps
        ctx.shapes = (mat.shape, vec.shape)
        ctx.dtypes = (mat.dtype, vec.dtype)
        ctx.save_for_backward(mat, x)
        return x

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import sym
        mat, x = ctx.saved_tensors
        N = x.shape[-1]
        g = g.contiguous()
        # x = A^-1 v  =>  v_bar = A^-T g,  A_bar = -v_bar x^T
        gv = sym.sym_solve(_transposed(mat.to(g.dtype), ctx.kind, N), g, eps=ctx.eps)
        g_mat = g_vec = None
        if ctx.needs_input_grad[0]:
            g_mat = _sum_to(_mat_cotangent(ctx.kind, gv, x, True), ctx.shapes[0]).to(ctx.dtypes[0])
        if ctx.needs_input_grad[1]:
            g_vec = _sum_to(gv, ctx.shapes[1]).to(ctx.dtypes[1])
        return g_mat, g_vec, None, None


class EigSymFn(torch.autograd.Function):
    """Giles, "An extended collection of matrix derivative results" (2008), sec. 3.1:
    A_bar = U (diag(D_bar) + F o (U^T U_bar)) U^T, F_ij = 1 / (d_j - d_i), F_ii = 0."""

    @staticmethod
    def forward(ctx, a, compute_u, upper, max_iter, tol):
        from . import qr
        val, vec = qr.eig_sym(a, compute_u=True, upper=upper, check_finite=False, max_iter=max_iter, tol=tol)
        val, vec = val.contiguous(), vec.contiguous()
        ctx.save_for_backward(val, vec)
        ctx.set_materialize_grads(False)
        if compute_u:
            return val, vec
        return val

    @staticmethod
    @once_differentiable
    def backward(ctx, gD, gU=None):
        from . import sym
        D, U = ctx.saved_tensors
        if gD is None and gU is None:
            return (None,) * 5
        n = D.shape[-1]
        if gU is None:
            # U diag(gD) U^T = J^T H J with J = U^T, H = diag(gD): one compact kernel + expansion
            return sym.sym_to_full(sym.sym_matmul(U.transpose(-1, -2), gD.contiguous())), None, None, None, None
        # eigenvector term (small dense products per matrix; element-wise glue in torch)
        F = D.unsqueeze(-2) - D.unsqueeze(-1)                     # F_ij = d_j - d_i
        F = torch.where(F == 0, torch.zeros_like(F), 1 / F)
        inner = F * _small_matmul(U.transpose(-1, -2), gU)
        if gD is not None:
            inner = inner + torch.diag_embed(gD)
        return _small_matmul(_small_matmul(U, inner), U.transpose(-1, -2)), None, None, None, None


def _small_matmul(a, b):
    """(..., n, n) @ (..., n, n) for tiny n as broadcast multiply-adds (rocBLAS batched GEMM is
    slow, and faults, on batches of ~1e7 tiny matrices)."""
    n = a.shape[-1]
    out = a[..., :, 0:1] * b[..., 0:1, :]
    for k in range(1, n):
        out = out + a[..., :, k:k + 1] * b[..., k:k + 1, :]
    return out


class SumFn(torch.autograd.Function):
    """sum / nansum / mean / nanmean: forward = one streaming kernel, backward = mask + broadcast."""

    @staticmethod
    def forward(ctx, input, dim, keepdim, omitnan, mean, dtype):
        from . import reduce as R
        if mean:
            out = R.mean(input, dim, keepdim, omitnan, dtype=dtype)
        else:
            out = R.sum(input, dim, keepdim, omitnan, dtype=dtype)
        ctx.cfg = (dim, keepdim, omitnan, mean)
        ctx.in_dtype = input.dtype
        ctx.save_for_backward(input)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import reduce as R
        from .utils import ensure_list
        (x,) = ctx.saved_tensors
        dim, keepdim, omitnan, mean = ctx.cfg
        nd = x.dim()
        if dim is None:
            dims = list(range(nd))
        else:
            dims = [d if d >= 0 else nd + d for d in ensure_list(dim)]
        if not keepdim:
            for d in sorted(dims):
                g = g.unsqueeze(d)
        if mean:
            if omitnan:
                w = R._reduce(_lib.RED_NANCOUNT, x, dim, True, torch.float64)[0]
                g = g / w.to(g.dtype)
            else:
                cnt = 1
                for d in dims:
                    cnt *= x.shape[d]
                g = g / cnt
        g = g.expand(x.shape)
        if omitnan:
            g = torch.where(torch.isnan(x), torch.zeros_like(g), g)
        return g.to(ctx.in_dtype), None, None, None, None, None
