"""Autograd for the facade: `torch.autograd.Function`s whose forward AND backward run the
HIP kernels of libnfm_hip.so (the backward of a solve is a solve, of a mat-vec a mat-vec,
plus one `nfm_sym_outer2` launch for the compact-matrix cotangent).  Element-wise glue in
the backward of the reductions (mask, broadcast, divide) is plain torch on the device.

Covered: sym_matvec / sym_addmatvec / sym_submatvec, sym_solve, sym_invert (+ diag), sym_det,
sym_to_full, sym_outer, sym_matmul,
batchmatvec / batchinv / batchdet, eig_sym (eigenvalues, and eigenvectors through Giles' formula,
as upstream's `_EigSym` `_impl/qr.py:684-735` intends), sum / nansum / mean / nanmean,
max / min / nanmax / nanmin (the cotangent goes to the selected element), var / std / nanvar /
nanstd.  Everything else is forward-only and says so.
"""
import ctypes
import torch
from torch.autograd.function import once_differentiable
from . import _lib
from ._dispatch import on_device, Batch, dtype_code, expand_batch, stream_ptr, broadcast_shapes


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
    with on_device(dev):
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
    def forward(ctx, a, compute_u, upper, max_iter, tol, arithmetic='reference'):
        from . import qr
        val, vec = qr.eig_sym(a, compute_u=True, upper=upper, check_finite=False, max_iter=max_iter, tol=tol,
                              arithmetic=arithmetic)
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
            return (None,) * 6
        n = D.shape[-1]
        if gU is None:
            # U diag(gD) U^T = J^T H J with J = U^T, H = diag(gD): one compact kernel + expansion
            return sym.sym_to_full(sym.sym_matmul(U.transpose(-1, -2), gD.contiguous())), None, None, None, None, None
        # eigenvector term (small dense products per matrix; element-wise glue in torch)
        F = D.unsqueeze(-2) - D.unsqueeze(-1)                     # F_ij = d_j - d_i
        F = torch.where(F == 0, torch.zeros_like(F), 1 / F)
        inner = F * _small_matmul(U.transpose(-1, -2), gU)
        if gD is not None:
            inner = inner + torch.diag_embed(gD)
        return _small_matmul(_small_matmul(U, inner), U.transpose(-1, -2)), None, None, None, None, None


def _small_matmul(a, b):
    """(..., n, m) @ (..., m, p) for tiny sizes as broadcast multiply-adds (rocBLAS batched GEMM is
    slow, and faults, on batches of ~1e7 tiny matrices)."""
    m = a.shape[-1]
    out = a[..., :, 0:1] * b[..., 0:1, :]
    for k in range(1, m):
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


# ---------------------------------------------------------------- batched / sym inverses, dets
def _unsq(g, nd_out, dims, keepdim):
    if not keepdim:
        for d in sorted(dims):
            g = g.unsqueeze(d)
    return g


class BatchMatvecFn(torch.autograd.Function):
    """y = A v:  dA = g v^T (broadcast product),  dv = A^T g (the same kernel on the transposed view)."""

    @staticmethod
    def forward(ctx, mat, vec):
        from . import batched as B
        ctx.save_for_backward(mat, vec)
        with torch.no_grad():
            return B.batchmatvec(mat, vec)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import batched as B
        mat, vec = ctx.saved_tensors
        gm = gv = None
        if ctx.needs_input_grad[0]:
            gm = _sum_to(g.unsqueeze(-1) * vec.unsqueeze(-2), mat.shape).to(mat.dtype)
        if ctx.needs_input_grad[1]:
            gv = _sum_to(B.batchmatvec(mat.transpose(-1, -2), g), vec.shape).to(vec.dtype)
        return gm, gv


class BatchInvFn(torch.autograd.Function):
    """B = A^-1:  dA = -B^T G B^T."""

    @staticmethod
    def forward(ctx, a, perturb):
        from . import batched as B
        with torch.no_grad():
            inv = B.batchinv(a, perturb=perturb)
        ctx.save_for_backward(inv)
        return inv

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        it = inv.transpose(-1, -2)
        return -_small_matmul(_small_matmul(it, g), it), None


class BatchDetFn(torch.autograd.Function):
    """d = det A:  dA = g d A^-T (one inverse kernel)."""

    @staticmethod
    def forward(ctx, a):
        from . import batched as B
        with torch.no_grad():
            d = B.batchdet(a)
        ctx.save_for_backward(a, d)
        return d

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import batched as B
        a, d = ctx.saved_tensors
        return (g * d).unsqueeze(-1).unsqueeze(-1) * B.batchinv(a).transpose(-1, -2)


def _full_to_compact_grad(gf):
    """pull a full-matrix cotangent (..., M, M) back onto compact storage: g_ii, g_ij + g_ji."""
    M = gf.shape[-1]
    parts = [gf.diagonal(dim1=-2, dim2=-1)]
    s = gf + gf.transpose(-1, -2)
    for i in range(M):
        if i + 1 < M:
            parts.append(s[..., i, i + 1:])
    return torch.cat(parts, -1)


class SymInvertFn(torch.autograd.Function):
    """compact B = A^-1 (or its diagonal): full cotangent -B G B pulled back onto compact storage."""

    @staticmethod
    def forward(ctx, mat, diag, dtype):
        from . import sym as S
        with torch.no_grad():
            inv = S.sym_invert(mat, dtype=dtype)
        ctx.save_for_backward(inv)
        ctx.diag = diag
        ctx.in_dtype = mat.dtype
        M = S._nb_prm(mat.shape[-1])
        return inv[..., :M].clone() if diag else inv

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import sym as S
        (inv,) = ctx.saved_tensors
        Bf = S.sym_to_full(inv)
        Gf = torch.diag_embed(g) if ctx.diag else S.sym_to_full(g)
        if not ctx.diag:
            # the compact cotangent counts an off-diagonal entry once: split it over (i, j) and (j, i)
            Gf = (Gf + torch.diag_embed(Gf.diagonal(dim1=-2, dim2=-1))) / 2
        full = -_small_matmul(_small_matmul(Bf, Gf), Bf)
        return _full_to_compact_grad(full).to(ctx.in_dtype), None, None


class SymDetFn(torch.autograd.Function):
    """d = det A (compact):  full cotangent g d A^-1 pulled back onto compact storage."""

    @staticmethod
    def forward(ctx, mat, dtype):
        from . import sym as S
        with torch.no_grad():
            d = S.sym_det(mat, dtype=dtype)
        ctx.save_for_backward(mat, d)
        ctx.in_dtype = mat.dtype
        return d

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import sym as S
        mat, d = ctx.saved_tensors
        inv = S.sym_invert(mat, dtype=d.dtype)          # compact A^-1
        M = S._nb_prm(mat.shape[-1])
        scale = torch.cat([torch.ones(M, dtype=d.dtype, device=d.device),
                           torch.full((inv.shape[-1] - M,), 2.0, dtype=d.dtype, device=d.device)])
        return ((g * d).unsqueeze(-1) * inv * scale).to(ctx.in_dtype), None


# ---------------------------------------------------------------- max / min, var / std
class PickFn(torch.autograd.Function):
    """max / min / nanmax / nanmin over dims: the cotangent goes to the selected element."""

    @staticmethod
    def forward(ctx, input, which, dim, keepdim, omitnan):
        from . import reduce as R
        fn = R.max if which == 'max' else R.min
        with torch.no_grad():
            if dim is None:
                flat = input.reshape(-1)
                val, idx = fn(flat, dim=0, omitnan=omitnan, return_indices=True)
                ctx.cfg = (None, keepdim, tuple(input.shape))
            else:
                val, idx = fn(input, dim=dim, keepdim=True, omitnan=omitnan, return_indices=True)
                ctx.cfg = (dim, keepdim, tuple(input.shape))
        ctx.save_for_backward(idx)
        ctx.in_dtype = input.dtype
        if dim is None:
            return val.reshape([1] * input.dim()) if keepdim else val
        if keepdim:
            return val
        nd = input.dim()
        from .utils import ensure_list
        dims = [d if d >= 0 else nd + d for d in ensure_list(dim)]
        return val.squeeze(tuple(dims))

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from .utils import ensure_list
        (idx,) = ctx.saved_tensors
        dim, keepdim, shape = ctx.cfg
        nd = len(shape)
        if dim is None:
            out = torch.zeros(shape, dtype=g.dtype, device=g.device)
            out.view(-1)[idx] = g.reshape(())
            return out.to(ctx.in_dtype), None, None, None, None
        dims = [d if d >= 0 else nd + d for d in ensure_list(dim)]
        g = g if keepdim else _unsq(g, nd, dims, False)
        scalar = not isinstance(dim, (list, tuple, range))
        sub = idx.unsqueeze(-1) if scalar else idx          # (keptshape..., len(dims))
        out = torch.zeros(shape, dtype=g.dtype, device=g.device)
        grids = torch.meshgrid(*[torch.arange(1 if d in dims else s, device=g.device) for d, s in enumerate(shape)],
                               indexing='ij')
        index = [sub[..., dims.index(d)] if d in dims else grids[d] for d in range(nd)]
        out.index_put_(tuple(index), g.expand(sub.shape[:-1]), accumulate=True)
        return out.to(ctx.in_dtype), None, None, None, None


class VarFn(torch.autograd.Function):
    """var / std (+ nan variants): d/dx = 2 (x - mean) / (n - ddof) [/ (2 std)] on the counted elements."""

    @staticmethod
    def forward(ctx, input, dim, keepdim, unbiased, omitnan, std, dtype):
        from . import reduce as R
        with torch.no_grad():
            fn = R.std if std else R.var
            out = fn(input, dim, True, unbiased, omitnan, dtype=torch.float64)
            mean = R.mean(input, dim, True, omitnan, dtype=torch.float64)
            if omitnan:
                cnt = R._reduce(_lib.RED_NANCOUNT, input, dim, True, torch.float64)[0]
            else:
                cnt = None
        ctx.save_for_backward(input, out, mean, cnt if cnt is not None else out)
        ctx.cfg = (dim, keepdim, unbiased, omitnan, std, cnt is not None)
        odt = dtype or input.dtype
        res = out.to(odt)
        if keepdim:
            return res
        nd = input.dim()
        from .utils import ensure_list
        dims = list(range(nd)) if dim is None else [d if d >= 0 else nd + d for d in ensure_list(dim)]
        return res.squeeze(tuple(dims)) if dims else res

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from .utils import ensure_list
        x, out, mean, cnt = ctx.saved_tensors
        dim, keepdim, unbiased, omitnan, std, has_cnt = ctx.cfg
        nd = x.dim()
        dims = list(range(nd)) if dim is None else [d if d >= 0 else nd + d for d in ensure_list(dim)]
        g = g.to(torch.float64)
        if not keepdim:
            g = _unsq(g, nd, dims, False)
        if has_cnt:
            n = cnt
        else:
            n = 1
            for d in dims:
                n *= x.shape[d]
        den = n - (1 if unbiased else 0)
        gx = 2.0 * (x.to(torch.float64) - mean) / den
        if std:
            gx = gx / (2.0 * out)
        gx = gx * g
        if omitnan:
            gx = torch.where(torch.isnan(x), torch.zeros_like(gx), gx)
        return gx.to(x.dtype), None, None, None, None, None, None


# ---------------------------------------------------------------- sym_to_full / sym_outer / sym_matmul
def _halved_full(g):
    """full symmetric matrix G with <G, full(c)> = <g, c> for every compact c: off-diagonals halved"""
    from . import sym as S
    Gf = S.sym_to_full(g)
    return (Gf + torch.diag_embed(Gf.diagonal(dim1=-2, dim2=-1))) / 2


class SymToFullFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mat, dtype):
        from . import sym as S
        ctx.in_dtype = mat.dtype
        with torch.no_grad():
            return S.sym_to_full(mat, dtype=dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return _full_to_compact_grad(g).to(ctx.in_dtype), None


class SymOuterFn(torch.autograd.Function):
    """compact(x x^T): dx_i = 2 g_ii x_i + sum_{j != i} g_ij x_j = one compact mat-vec."""

    @staticmethod
    def forward(ctx, x, dtype):
        from . import sym as S
        ctx.save_for_backward(x)
        with torch.no_grad():
            return S.sym_outer(x, dtype=dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import sym as S
        (x,) = ctx.saved_tensors
        M = x.shape[-1]
        g2 = torch.cat([2 * g[..., :M], g[..., M:]], -1)
        return S.sym_matvec(g2, x.to(g.dtype)).to(x.dtype), None


class SymMatmulFn(torch.autograd.Function):
    """J^T H J (or J H J^T where the reference computes that, quirk Q16), compact H and output."""

    @staticmethod
    def forward(ctx, j, h, dtype):
        from . import sym as S
        k, d = j.shape[-2:]
        ctx.sym = h.shape[-1] == k * (k + 1) // 2
        ctx.flip = ctx.sym and k == d and k in (2, 3)
        ctx.save_for_backward(j, h)
        with torch.no_grad():
            return S.sym_matmul(j, h, dtype=dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import sym as S
        j, h = ctx.saved_tensors
        Gf = _halved_full(g)
        J = j.to(g.dtype)
        Hf = S.sym_to_full(h.to(g.dtype)) if ctx.sym else torch.diag_embed(h.to(g.dtype))
        gj = gh = None
        if ctx.flip:     # out = J H J^T
            if ctx.needs_input_grad[0]:
                gj = 2 * _small_matmul(_small_matmul(Gf, J), Hf)
            if ctx.needs_input_grad[1]:
                gh = _small_matmul(_small_matmul(J.transpose(-1, -2), Gf), J)
        else:            # out = J^T H J
            if ctx.needs_input_grad[0]:
                gj = 2 * _small_matmul(_small_matmul(Hf, J), Gf)
            if ctx.needs_input_grad[1]:
                gh = _small_matmul(_small_matmul(J, Gf), J.transpose(-1, -2))
        if gh is not None:
            gh = _full_to_compact_grad(gh) if ctx.sym else gh.diagonal(dim1=-2, dim2=-1)
            gh = _sum_to(gh, h.shape).to(h.dtype)
        if gj is not None:
            gj = _sum_to(gj, j.shape).to(j.dtype)
        return gj, gh, None
