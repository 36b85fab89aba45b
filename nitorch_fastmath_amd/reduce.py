r'''
NaN-aware reductions on MI355X -- drop-in for `nitorch_fastmath.reduce`
(`reduce.py:38-41`): same functions, same argument order and defaults.

- all functions can reduce across multiple dimensions simultaneously;
- min/max only return the reduced tensor by default; `return_indices=True` also
  returns the indices of the picked elements, shaped `(..., len(dim))` (the last
  axis is dropped for a scalar `dim`);
- all functions have an `omitnan` argument, or a `nan*` version where it is implied.

The reference makes several full passes over memory per call (clone, isnan,
masked_fill, reduce: `reduce.py:502-510`) and half of it raises on current PyTorch
(SURVEY quirks Q10-Q13).  Here every reduction is one streaming HIP kernel (NaN ->
identity by select, 64-lane wavefront shuffle reduce, double accumulation), `median` a radix
selection, and the functions that raise upstream implement their documented semantics.

`inplace=True` only ever meant "the input MAY be modified" (`reduce.py:72-74`); the reference
uses that permission (its `nansum` zeroes, its `nanmax` / `nanmin` write -inf / +inf over, the
NaNs of the caller's tensor: `reduce.py:502-509`, `:258-260`), this backend never needs to: the
flag is accepted and the input is left bit-identical (pinned by
tests/test_gpu_reduce.py::test_inplace_never_modifies_the_input, recorded in INTEGRATION.md).
'''
__all__ = [
    'min', 'max', 'nanmin', 'nanmax', 'median',
    'sum', 'nansum', 'mean', 'nanmean', 'var', 'nanvar', 'std', 'nanstd'
]
import builtins
import torch
from . import _lib
from ._dispatch import on_device, dtype_code, no_grad_required, require_gpu, stream_ptr
from .utils import ensure_list, ind2sub


def _needs_grad(t):
    return torch.is_grad_enabled() and t.requires_grad


def _SumFn():
    from ._autograd import SumFn
    return SumFn


def _prod(xs):
    p = 1
    for x in xs:
        p *= int(x)
    return p


def _workspace(dev, nbytes=None):
    n = _lib.lib().nfm_reduce_workspace_bytes() if nbytes is None else int(nbytes)
    if n == 0:
        return None, 0
    return torch.empty(n, dtype=torch.uint8, device=dev), n


def _view3(input, dim):
    """Contiguous (outer, red, inner) view of `input` for a reduction over `dim`.

    Consecutive dims of a contiguous tensor are reduced in place; anything else is first
    permuted so that the reduced dims are last (one copy).
    Returns (x, outer, red, inner, dims, kept, redshape)."""
    nd = input.dim()
    if dim is None:
        dims = list(range(nd))
    else:
        dims = [d if d >= 0 else nd + d for d in ensure_list(dim)]
    if len(set(dims)) != len(dims) or builtins.min(dims, default=0) < 0 or builtins.max(dims, default=-1) >= nd:
        raise IndexError(f'invalid reduction dims {dim} for a {nd}-d tensor')
    shape = list(input.shape)
    kept = [d for d in range(nd) if d not in dims]
    redshape = [shape[d] for d in dims]
    consecutive = len(dims) > 0 and dims == list(range(dims[0], dims[0] + len(dims)))
    if nd == 0 or len(dims) == 0:
        x, outer, red, inner = input.contiguous().reshape(-1), input.numel(), 1, 1
    elif consecutive and input.is_contiguous():
        x = input
        outer, red, inner = _prod(shape[:dims[0]]), _prod(redshape), _prod(shape[dims[-1] + 1:])
    else:
        x = input.permute(kept + dims).contiguous()
        outer, red, inner = _prod([shape[d] for d in kept]), _prod(redshape), 1
    return x, outer, red, inner, dims, kept, redshape


def _canon(input, dim):
    """A non-contiguous tensor that is a dim permutation of a contiguous one (e.g. the
    channel-last view of a channel-first field): return (contiguous view, mapped dims,
    inverse permutation) so that the reduction runs in place instead of on a copy."""
    if dim is None or input.is_contiguous() or input.dim() < 2:
        return None
    nd = input.dim()
    st = input.stride()
    perm = sorted(range(nd), key=lambda d: (-st[d], d))
    xp = input.permute(perm)
    if not xp.is_contiguous():
        return None
    inv = [0] * nd
    for j, d in enumerate(perm):
        inv[d] = j
    scalar = not isinstance(dim, (list, tuple, range))
    dims = [d if d >= 0 else nd + d for d in ensure_list(dim)]
    if any(d < 0 or d >= nd for d in dims):
        return None
    mapped = [inv[d] for d in dims]
    return xp, (mapped[0] if scalar else mapped), inv, dims


def _uncanon(r, inv, dims, keepdim):
    r = r.permute(inv)
    return r if keepdim or not dims else r.squeeze(tuple(dims))


def _dim_groups(input, dim):
    """Sorted reduction dims split into runs of consecutive dims, or None when one kernel
    (or the permuting copy) handles the request anyway."""
    if dim is None or not input.is_contiguous():
        return None
    nd = input.dim()
    dims = sorted({d if d >= 0 else nd + d for d in ensure_list(dim)})
    if len(dims) != len(ensure_list(dim)) or (dims and (dims[0] < 0 or dims[-1] >= nd)):
        return None   # duplicates / out of range: let the one-kernel path raise
    groups = []
    for d in dims:
        if groups and d == groups[-1][-1] + 1:
            groups[-1].append(d)
        else:
            groups.append([d])
    return groups if len(groups) > 1 else None


# what a later stage does with the partial results of an earlier one
_STAGE2 = {_lib.RED_NANSUM: _lib.RED_NANSUM, _lib.RED_SUM: _lib.RED_SUM, _lib.RED_NANCOUNT: _lib.RED_SUM,
           _lib.RED_NANSUMSQ: _lib.RED_SUM, _lib.RED_NANMAX: _lib.RED_NANMAX, _lib.RED_NANMIN: _lib.RED_NANMIN,
           _lib.RED_MAX: _lib.RED_MAX, _lib.RED_MIN: _lib.RED_MIN}


def _reduce_staged(op, input, groups, keepdim, out_dtype):
    """Non-adjacent reduction dims (e.g. batch + spatial dims of a channel-first field): reduce
    the runs of adjacent dims one after the other, last run first, instead of permuting the
    whole tensor into a copy.  Sums travel between the stages in float64."""
    sums = op in (_lib.RED_NANSUM, _lib.RED_SUM, _lib.RED_NANCOUNT, _lib.RED_NANSUMSQ)
    y, stage_op = input, op
    for g in reversed(groups):
        last = g is groups[0]
        y = _reduce(stage_op, y, g, False, out_dtype if last else (torch.float64 if sums else input.dtype))[0]
        stage_op = _STAGE2[op]
    if keepdim:
        red = {d for g in groups for d in g}
        y = y.reshape([1 if d in red else s for d, s in enumerate(input.shape)])
    return y


def _reduce(op, input, dim, keepdim, out_dtype, want_idx=False):
    """Run one reduction kernel.  Returns (values, flat_indices or None, dims, redshape)."""
    input = torch.as_tensor(input)
    dev = require_gpu(input)
    no_grad_required(input)
    canon = _canon(input, dim)
    if canon is not None:
        xp, mapped, inv, odims = canon
        val, idx, dims, redshape = _reduce(op, xp, mapped, True, out_dtype, want_idx)
        val = _uncanon(val, inv, odims, keepdim)
        if idx is not None:
            idx = _uncanon(idx, inv, odims, keepdim)
        return val, idx, dims, redshape
    if not want_idx:
        groups = _dim_groups(input, dim)
        if groups is not None:
            return _reduce_staged(op, input, groups, keepdim, out_dtype), None, None, None
    code = dtype_code(input.dtype)
    ocode = dtype_code(out_dtype)
    L = _lib.lib()
    nd = input.dim()
    if dim is None:
        x = input if input.is_contiguous() else input.contiguous()
        out = torch.empty([], dtype=out_dtype, device=dev)
        ws, wsn = _workspace(dev)
        with on_device(dev):
            _lib.check(L.nfm_reduce_all(code, op, ocode, x.numel(), x.data_ptr(), ws.data_ptr(), wsn,
                                        out.data_ptr(), stream_ptr(dev)))
        if keepdim:
            out = out.reshape([1] * nd)
        return out, None, None, None
    x, outer, red, inner, dims, kept, redshape = _view3(input, dim)
    shape = list(input.shape)
    subshape = [shape[d] for d in kept]
    out = torch.empty(subshape, dtype=out_dtype, device=dev)
    idx = torch.empty(subshape, dtype=torch.long, device=dev) if want_idx else None
    if red == 0 and op in (_lib.RED_NANMAX, _lib.RED_NANMIN, _lib.RED_MAX, _lib.RED_MIN):
        raise IndexError('cannot take the max/min over an empty dimension')
    ws, wsn = _workspace(dev, L.nfm_reduce_dim_workspace_bytes(code, op, outer, red, inner, int(want_idx)))
    with on_device(dev):
        _lib.check(L.nfm_reduce_dim(code, op, ocode, outer, red, inner, x.data_ptr(),
                                    ws.data_ptr() if ws is not None else None, wsn, out.data_ptr(),
                                    idx.data_ptr() if idx is not None else None, stream_ptr(dev)))
    if keepdim:
        keptshape = [1 if d in dims else s for d, s in enumerate(shape)]
        out = out.reshape(keptshape)
        if idx is not None:
            idx = idx.reshape(keptshape)
    return out, idx, dims, redshape


def _deliver(val, out):
    if out is None:
        return val
    if tuple(out.shape) != tuple(val.shape):
        out.resize_(val.shape)
    out.copy_(val)
    return out


def _reduce_index(op_plain, op_nan, input, dim, keepdim, omitnan, return_indices, out):
    """max/min driver: semantics of `_reduce_index` (`reduce.py:49-142`)."""
    input = torch.as_tensor(input)
    op = op_nan if omitnan else op_plain
    out_val, out_ind = ensure_list(out, 2, default=None) if out is not None else (None, None)
    if dim is None:
        if input.numel() == 0:
            raise RuntimeError('max/min of an empty tensor')
        val, _, _, _ = _reduce(op, input, None, False, input.dtype)
        return _deliver(val, out_val)
    scalar_dim = not isinstance(dim, (list, tuple, range))
    val, idx, dims, redshape = _reduce(op, input, dim, keepdim, input.dtype, want_idx=return_indices)
    val = _deliver(val, out_val)
    if not return_indices:
        return val
    if len(redshape) == 1:                # one reduced dim: the position along it IS the sub-index
        sub = idx if scalar_dim else idx.unsqueeze(-1)
    else:
        sub = ind2sub(idx, redshape)          # (len(dim), ...)
        sub = torch.movedim(sub, 0, -1)       # (..., len(dim))
        if scalar_dim:
            sub = sub[..., 0]
    return val, _deliver(sub, out_ind)


def _pick_with_grad(which, input, dim, keepdim, omitnan, return_indices, out):
    """differentiable max / min: values through `PickFn`, indices (no gradient) on the side"""
    from ._autograd import PickFn
    if out is not None:
        raise RuntimeError('out= is not supported for tensors that require grad')
    input = torch.as_tensor(input)
    val = PickFn.apply(input, which, dim, keepdim, omitnan)
    if not return_indices or dim is None:
        return val
    with torch.no_grad():
        _, idx = (max if which == 'max' else min)(input.detach(), dim, keepdim, omitnan, False, True)
    return val, idx


def max(input, dim=None, keepdim=False, omitnan=False, inplace=False, return_indices=False, out=None):
    r"""Multi-dimensional max reduction (`reduce.py:145-197`).

    max(input) -> Tensor; max(input, dim) -> Tensor;
    max(input, dim, return_indices=True) -> (Tensor, Tensor)
    """
    if _needs_grad(torch.as_tensor(input)):
        return _pick_with_grad('max', input, dim, keepdim, omitnan, return_indices, out)
    return _reduce_index(_lib.RED_MAX, _lib.RED_NANMAX, input, dim, keepdim, omitnan, return_indices, out)


def min(input, dim=None, keepdim=False, omitnan=False, inplace=False, return_indices=False, out=None):
    r"""Multi-dimensional min reduction (`reduce.py:200-252`)."""
    if _needs_grad(torch.as_tensor(input)):
        return _pick_with_grad('min', input, dim, keepdim, omitnan, return_indices, out)
    return _reduce_index(_lib.RED_MIN, _lib.RED_NANMIN, input, dim, keepdim, omitnan, return_indices, out)


def nanmax(input, dim=None, keepdim=False, inplace=False, return_indices=False, out=None):
    r"""Multi-dimensional max reduction, excluding NaNs (`reduce.py:267-316`); all-NaN -> -inf."""
    return max(input, dim, keepdim, True, inplace, return_indices, out)


def nanmin(input, dim=None, keepdim=False, inplace=False, return_indices=False, out=None):
    r"""Multi-dimensional min reduction, excluding NaNs (`reduce.py:331-380`); all-NaN -> +inf."""
    return min(input, dim, keepdim, True, inplace, return_indices, out)


def median(input, dim=None, keepdim=False, omitnan=False, inplace=False, return_indices=False, out=None):
    r"""Multi-dimensional median (`reduce.py:384-428`): the lower of the two middle values for an
    even count, and the index of the first element that holds it.

    `omitnan=False` (the default) is what the reference computes: it calls `torch.median`, so a NaN
    in a slice makes that slice's median NaN -- its docstring says "always omits NaNs" but its code
    never does (pinned by tests/test_gpu_reduce.py::test_median_semantics).
    `omitnan=True` is the documented intent: the median of the non-NaN values (all-NaN -> NaN).

    One kernel (`nfm_reduce_median`: a register sorting network per row for short rows, radix
    selection for long ones), no copy of the data beyond the move of the reduced dims to the end that
    the reference does too (`reduce.py:112-113`) -- and not even that one for short rows reduced in
    the middle of a contiguous tensor (`nfm_reduce_median_mid`).
    """
    input = torch.as_tensor(input)
    dev = require_gpu(input)
    grad = _needs_grad(input)
    if grad and out is not None:
        raise RuntimeError('out= is not supported for tensors that require grad')
    code = dtype_code(input.dtype)
    nd = input.dim()
    if not grad and out is None and isinstance(dim, int) and not input.is_contiguous():
        # a dim permutation of a contiguous tensor (the channel-last VIEW of a channel-first field):
        # reduce the corresponding dim of the contiguous tensor in place instead of copying the view
        canon = _canon(input, dim)
        if canon is not None:
            xp, mapped, inv, odims = canon
            r = median(xp, mapped, True, omitnan, False, return_indices)
            if return_indices:
                return _uncanon(r[0], inv, odims, keepdim), _uncanon(r[1], inv, odims, keepdim)
            return _uncanon(r, inv, odims, keepdim)
    scalar_dim = dim is not None and not isinstance(dim, (list, tuple, range))
    dims = list(range(nd)) if dim is None else [d if d >= 0 else nd + d for d in ensure_list(dim)]
    if len(set(dims)) != len(dims) or any(d < 0 or d >= nd for d in dims):
        raise IndexError(f'invalid reduction dims {dim} for a {nd}-d tensor')
    kept = [d for d in range(nd) if d not in dims]
    redshape = [input.shape[d] for d in dims]
    subshape = [input.shape[d] for d in kept]
    red, rows = _prod(redshape), _prod(subshape)
    if red == 0:
        raise IndexError('cannot take the median over an empty dimension')
    val = torch.empty(rows, dtype=input.dtype, device=dev)
    want_idx = return_indices and dim is not None
    idx = torch.empty(rows, dtype=torch.long, device=dev) if (want_idx or grad) else None
    L = _lib.lib()
    # A block of adjacent dims reduced in the MIDDLE of a contiguous tensor (the channel dim of a
    # channel-first field): short rows are sorted one per lane straight from the (outer, red, inner)
    # layout -- no transposing copy of the whole tensor first (upstream makes one, `reduce.py:112-113`).
    d0 = builtins.min(dims)
    inner = _prod(input.shape[d0 + len(dims):])
    if (dims == list(range(d0, d0 + len(dims))) and inner > 1 and input.is_contiguous() and rows >= 4096
            and 2 <= red <= L.nfm_reduce_median_lane_max(code)):
        with on_device(dev):
            _lib.check(L.nfm_reduce_median_mid(code, int(bool(omitnan)), rows // inner, red, inner,
                                               input.detach().data_ptr(), val.data_ptr(),
                                               idx.data_ptr() if idx is not None else None, stream_ptr(dev)))
        rows_done = True
    else:
        rows_done = False
    x = None
    if grad or not rows_done:
        xg = input.permute(kept + dims).reshape(rows, red)   # differentiable view / copy of the rows
    if not rows_done:
        x = xg.detach()
        if not x.is_contiguous():
            x = x.contiguous()
    step = rows if red <= 1024 else 16384          # long rows: grid.y bound of the histogram passes; 16 KiB of 64-bit bins per row
    for lo in ([] if rows_done else range(0, rows, builtins.max(step, 1))):
        hi = builtins.min(rows, lo + step)
        ws, wsn = _workspace(dev, L.nfm_reduce_median_workspace_bytes(hi - lo, red))
        with on_device(dev):
            _lib.check(L.nfm_reduce_median(code, int(bool(omitnan)), hi - lo, red, x[lo:hi].data_ptr(),
                                           ws.data_ptr() if ws is not None else None, wsn, val[lo:hi].data_ptr(),
                                           idx[lo:hi].data_ptr() if idx is not None else None, stream_ptr(dev)))
    if grad:      # the gradient goes to the selected element: pick it out of the differentiable rows
        val = xg.gather(1, idx[:, None])[:, 0]
    shape = [1 if d in dims else s for d, s in enumerate(input.shape)] if keepdim else subshape
    val = val.reshape(shape)
    out_val, out_ind = ensure_list(out, 2, default=None) if out is not None else (None, None)
    val = _deliver(val, out_val)
    if not want_idx:
        return val
    if len(redshape) == 1:                # one reduced dim: the position along it IS the sub-index
        sub = idx.reshape(shape) if scalar_dim else idx.reshape(shape).unsqueeze(-1)
    else:
        sub = torch.movedim(ind2sub(idx.reshape(shape), redshape), 0, -1)
        if scalar_dim:
            sub = sub[..., 0]
    return val, _deliver(sub, out_ind)


def sum(input, dim=None, keepdim=False, omitnan=False, inplace=False, dtype=None, out=None):
    """Sum of a tensor (`reduce.py:431-468`); `dtype` is the accumulator/output dtype."""
    input = torch.as_tensor(input)
    if _needs_grad(input):
        return _deliver(_SumFn().apply(input, dim, keepdim, omitnan, False, dtype), out)
    op = _lib.RED_NANSUM if omitnan else _lib.RED_SUM
    val, _, _, _ = _reduce(op, input, dim, keepdim, dtype or input.dtype)
    return _deliver(val, out)


def nansum(input, dim=None, keepdim=False, inplace=False, dtype=None, out=None):
    """Sum of a tensor, excluding NaNs (`reduce.py:471-510`); all-NaN -> 0."""
    return sum(input, dim, keepdim, True, inplace, dtype, out)


def _moments(input, dim, keepdim):
    """One pass: per output entry [count, sum(x - K), sum((x - K)^2), K] of the non-NaN values."""
    input = torch.as_tensor(input)
    dev = require_gpu(input)
    no_grad_required(input)
    code = dtype_code(input.dtype)
    L = _lib.lib()
    x, outer, red, inner, dims, kept, _ = _view3(input, dim)
    shape = list(input.shape)
    subshape = [shape[d] for d in kept]
    out = torch.zeros(subshape + [4], dtype=torch.float64, device=dev)
    ws, wsn = _workspace(dev, L.nfm_reduce_moments_workspace_bytes(code, outer, red, inner))
    with on_device(dev):
        _lib.check(L.nfm_reduce_moments(code, outer, red, inner, x.data_ptr(),
                                        ws.data_ptr() if ws is not None else None, wsn,
                                        out.data_ptr(), stream_ptr(dev)))
    if keepdim:
        out = out.reshape([1 if d in dims else s for d, s in enumerate(shape)] + [4])
    return out[..., 0], out[..., 1], out[..., 2], out[..., 3], red


_STAT_MEAN, _STAT_VAR, _STAT_STD, _STAT_OMITNAN, _STAT_UNBIASED = 0, 1, 2, 4, 8


def _stat_staged(kind, input, groups, keepdim, omitnan, unbiased, out_dtype):
    """mean / var / std over non-adjacent dims: one pass of raw moments over the last run of
    adjacent dims, then the (small) per-slice moments are merged over the other dims with the
    pairwise formulas (count-weighted means, sum of within- and between-slice squares)."""
    last = groups[-1]
    n, s_, q, k, _ = _moments(input, last, False)
    rest = [d for g in groups[:-1] for d in g]
    has = n > 0
    safe_n = torch.where(has, n, torch.ones_like(n))
    mean_i = torch.where(has, k + s_ / safe_n, torch.zeros_like(n))
    m2_i = torch.where(has, q - s_ * s_ / safe_n, torch.zeros_like(n))
    cnt = n.sum(dim=rest)
    mean = (n * mean_i).sum(dim=rest) / cnt
    if kind == _STAT_MEAN:
        v = mean
    else:
        mexp = mean
        for d in rest:
            mexp = mexp.unsqueeze(d)
        m2 = (m2_i + n * (mean_i - mexp) ** 2).sum(dim=rest)
        den = cnt - 1 if unbiased else cnt
        v = m2.clamp_min(0) / den
        v = torch.where(den > 0, v, torch.full_like(v, float('nan')))
        if kind == _STAT_STD:
            v = v.sqrt()
    if not omitnan:
        total = 1
        for g in groups:
            for d in g:
                total *= input.shape[d]
        v = torch.where(cnt == total, v, torch.full_like(v, float('nan')))
    if keepdim:
        red = {d for g in groups for d in g}
        v = v.reshape([1 if d in red else sz for d, sz in enumerate(input.shape)])
    return v.to(out_dtype)


def _stat(kind, input, dim, keepdim, omitnan, unbiased, out_dtype):
    """mean / var / std in one pass (`nfm_reduce_stat`): moments finished inside the kernel."""
    input = torch.as_tensor(input)
    dev = require_gpu(input)
    no_grad_required(input)
    canon = _canon(input, dim)
    if canon is not None:
        xp, mapped, inv, odims = canon
        return _uncanon(_stat(kind, xp, mapped, True, omitnan, unbiased, out_dtype), inv, odims, keepdim)
    groups = _dim_groups(input, dim)
    if groups is not None:
        return _stat_staged(kind, input, groups, keepdim, omitnan, unbiased, out_dtype)
    code = dtype_code(input.dtype)
    ocode = dtype_code(out_dtype)
    L = _lib.lib()
    x, outer, red, inner, dims, kept, _ = _view3(input, dim)
    shape = list(input.shape)
    out = torch.empty([shape[d] for d in kept], dtype=out_dtype, device=dev)
    if out.numel() and red == 0:
        out.fill_(float('nan'))
    stat = kind | (_STAT_OMITNAN if omitnan else 0) | (_STAT_UNBIASED if unbiased else 0)
    ws, wsn = _workspace(dev, L.nfm_reduce_moments_workspace_bytes(code, outer, red, inner))
    with on_device(dev):
        _lib.check(L.nfm_reduce_stat(code, stat, ocode, outer, red, inner, x.data_ptr(),
                                     ws.data_ptr() if ws is not None else None, wsn,
                                     out.data_ptr(), stream_ptr(dev)))
    if keepdim:
        out = out.reshape([1 if d in dims else s for d, s in enumerate(shape)])
    return out


def mean(input, dim=None, keepdim=False, omitnan=False, inplace=False, dtype=None, out=None):
    """Mean of a tensor (`reduce.py:513-550`)."""
    input = torch.as_tensor(input)
    if _needs_grad(input):
        return _deliver(_SumFn().apply(input, dim, keepdim, omitnan, True, dtype), out)
    return _deliver(_stat(_STAT_MEAN, input, dim, keepdim, omitnan, False, dtype or input.dtype), out)


def nanmean(input, dim=None, keepdim=False, inplace=False, dtype=None, out=None):
    """Mean of a tensor, excluding NaNs (`reduce.py:553-594`; raises upstream, quirk Q11)."""
    return mean(input, dim, keepdim, True, inplace, dtype, out)


def var(input, dim=None, keepdim=False, unbiased=True, omitnan=False, inplace=False, dtype=None, out=None):
    """Variance of a tensor (`reduce.py:597-635`; the non-NaN form raises upstream, quirk Q13)."""
    input = torch.as_tensor(input)
    if _needs_grad(input):
        from ._autograd import VarFn
        return _deliver(VarFn.apply(input, dim, keepdim, unbiased, omitnan, False, dtype), out)
    return _deliver(_stat(_STAT_VAR, input, dim, keepdim, omitnan, unbiased, dtype or input.dtype), out)


def nanvar(input, dim=None, keepdim=False, unbiased=True, inplace=False, dtype=None, out=None):
    """Variance of a tensor, excluding NaNs (`reduce.py:638-685`; quirks Q11/Q12 fixed)."""
    return var(input, dim, keepdim, unbiased, True, inplace, dtype, out)


def std(input, dim=None, keepdim=False, unbiased=True, omitnan=False, inplace=False, dtype=None, out=None):
    """Standard deviation of a tensor (`reduce.py:688-726`)."""
    input = torch.as_tensor(input)
    if _needs_grad(input):
        from ._autograd import VarFn
        return _deliver(VarFn.apply(input, dim, keepdim, unbiased, omitnan, True, dtype), out)
    return _deliver(_stat(_STAT_STD, input, dim, keepdim, omitnan, unbiased, dtype or input.dtype), out)


def nanstd(input, dim=None, keepdim=False, unbiased=True, inplace=False, dtype=None, out=None):
    """Standard deviation of a tensor, excluding NaNs (`reduce.py:729-763`)."""
    return std(input, dim, keepdim, unbiased, True, inplace, dtype, out)
