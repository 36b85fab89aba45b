"""Host-side plumbing between torch tensors and the C ABI: dtype codes, stream,
broadcast normalisation and the two-level batch collapse (no data movement unless
the broadcast batch genuinely needs more than two stride levels)."""
import torch
from . import _lib

_DTYPES = {torch.float32: _lib.F32, torch.float64: _lib.F64}


def dtype_code(dtype):
    try:
        return _DTYPES[dtype]
    except KeyError:
        raise TypeError(f'nitorch_fastmath_amd supports float32 and float64 tensors, got {dtype}') from None


def require_gpu(*tensors):
    """The backend is HIP-only: refuse anything that is not on a ROCm/HIP device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                'nitorch_fastmath_amd runs on MI355X (HIP) tensors only; got a tensor on '
                f'{t.device}. There is no CPU fallback: move the data with .cuda().')
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f'all tensors must be on the same device ({dev} vs {t.device})')
    return dev


def no_grad_required(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(
            'nitorch_fastmath_amd kernels are forward-only (the reference documents that autograd '
            'does not work through these functions either); call under torch.no_grad() or detach().')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def stream_ptr(device):
    """raw hipStream_t of torch's current stream on `device` (the private accessor is 10x cheaper
    than building a torch.cuda.Stream object: small batches are launch-bound).  A private accessor that
    is gone or has changed its signature falls back to the public API; that it still MEANS the same is
    pinned by tests/test_gpu_streams_graphs.py::test_fast_accessors_agree_with_the_public_api."""
    global _raw_stream
    if _raw_stream is not None:
        try:
            return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
        except (TypeError, RuntimeError):
            _raw_stream = None
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """device guard for the C-ABI call, skipped when `device` already is the current device"""
    global _cur_device
    if _cur_device is not None and device.index is not None:
        try:
            if _cur_device() == device.index:
                return _NO_GUARD
        except (TypeError, RuntimeError):
            _cur_device = None
    return torch.cuda.device(device)


def same_dtype(tensors, dtype):
    """[t.to(dtype)] without the call when nothing changes"""
    return [t if (t is None or t.dtype == dtype) else t.to(dtype) for t in tensors]


def broadcast_shapes(*shapes):
    """`torch.broadcast_shapes` without its tracing machinery (12 us -> 1 us per call)."""
    first = shapes[0]
    if all(s == first for s in shapes[1:]):
        return torch.Size(first)
    nd = max(len(s) for s in shapes)
    out = [1] * nd
    for s in shapes:
        for k in range(1, len(s) + 1):
            d = s[-k]
            if d != 1:
                if out[-k] != 1 and out[-k] != d:
                    raise RuntimeError(f'Shape mismatch: objects cannot be broadcast to a single shape: {shapes}')
                out[-k] = d
    return torch.Size(out)


def common_dtype(dtype, *tensors):
    if dtype is not None:
        return dtype
    out = None
    for t in tensors:
        if t is None:
            continue
        out = t.dtype if out is None else torch.promote_types(out, t.dtype)
    return out


def _collapse(shape, strides_list):
    """Jointly merge adjacent batch dims that are mergeable for EVERY operand.
    Returns (sizes, [strides per operand])."""
    dims = [(s, [st[i] for st in strides_list]) for i, s in enumerate(shape) if s != 1]
    if not dims:
        return [], [[] for _ in strides_list]
    merged = [dims[0]]
    for s, sts in dims[1:]:
        ps, psts = merged[-1]
        if all(p == s * q for p, q in zip(psts, sts)):
            merged[-1] = (ps * s, sts)
        else:
            merged.append((s, sts))
    sizes = [m[0] for m in merged]
    per_op = [[m[1][k] for m in merged] for k in range(len(strides_list))]
    return sizes, per_op


class Batch:
    """Broadcast batch of several operands flattened to (n_outer, n_inner)."""

    def __init__(self, batch_shape, tensors, ncomp, _copyback=None, pack=False):
        # tensors[k] has shape batch_shape + comp dims (ncomp[k] trailing dims), already
        # expanded; the LAST tensor is the output.
        # pack=True (orders 9..16): the fast kernels of those orders want records back to back (or, since
        # round 3, component-major fields: `spd_strided_kernel` addresses any strides, and with the batch
        # running along memory its lanes read consecutive addresses).  Records that are strided ALONG the batch
        # (every other record, padded records, component stride 2) would be fetched lane by lane, 4 bytes at a
        # time, a record apart: measured 0.9-1.4 TB/s against 1.8-2.3 for one packing copy (2x the operand's
        # bytes at ~5 TB/s) + the fast kernel, so those are still packed; broadcast and component-major
        # operands are left alone (pack='all': component-major ones too, for the ops without such a kernel).
        nb = len(batch_shape)
        self.shape = tuple(batch_shape)
        self._copyback = _copyback
        # fast path (launch-bound small batches): every operand contiguous and of the full batch shape
        # -> one inner level, record strides straight from the component dims
        if all(t.is_contiguous() and tuple(t.shape[:nb]) == self.shape for t in tensors):
            numel = 1
            for sz in self.shape:
                numel *= sz
            if numel > 0:
                self.tensors = tensors
                self.n_outer, self.n_inner = 1, numel
                self.operands = []
                for t, nc in zip(tensors, ncomp):
                    cs = t.shape[nb:]
                    if nc == 2:
                        rec, sr, sc = cs[0] * cs[1], cs[1], 1
                    elif nc == 1:
                        rec, sr, sc = cs[0], 0, 1
                    else:
                        rec, sr, sc = 1, 0, 0
                    self.operands.append(_lib.Operand(t.data_ptr(), 0, rec if numel > 1 else 0, sr, sc))
                return
        if pack:
            tensors = self._pack(tensors, nb, keep_soa=(pack != 'all'))
        strides = [list(t.stride()[:nb]) for t in tensors]
        sizes, per_op = _collapse(self.shape, strides)
        if len(sizes) > 2:
            # more than two stride levels: materialise (facade-allocated outputs are
            # contiguous, so only exotic views pay this copy; a user-provided strided
            # `out=` is written through a temporary and copied back by finish())
            tensors = self._materialise(tensors)
            strides = [list(t.stride()[:nb]) for t in tensors]
            sizes, per_op = _collapse(self.shape, strides)
            assert len(sizes) <= 1, sizes
        self.tensors = tensors
        numel = 1
        for s in self.shape:
            numel *= s
        if numel == 0:
            self.n_outer, self.n_inner = 0, 0
            per_op = [[0, 0] for _ in tensors]
        elif len(sizes) == 0:
            self.n_outer, self.n_inner = 1, 1
            per_op = [[0, 0] for _ in tensors]
        elif len(sizes) == 1:
            self.n_outer, self.n_inner = 1, sizes[0]
            per_op = [[0, p[0]] for p in per_op]
        else:
            self.n_outer, self.n_inner = sizes
        if self.n_outer > 65535:
            # grid.y limit: fall back to one contiguous level
            tensors = self._materialise(tensors)
            self.__init__(batch_shape, tensors, ncomp, self._copyback)
            return
        self.operands = []
        for t, (so, si), nc in zip(tensors, per_op, ncomp):
            cs = t.stride()[nb:]
            if nc == 2:
                sr, sc = cs
            elif nc == 1:
                sr, sc = 0, cs[0]
            else:
                sr, sc = 0, 0
            self.operands.append(_lib.Operand(t.data_ptr(), so, si, sr, sc))


    def _pack(self, tensors, nb, keep_soa=True):
        new = []
        for k, t in enumerate(tensors):
            bcast = any(st == 0 and sz > 1 for st, sz in zip(t.stride()[:nb], t.shape[:nb]))
            # component-major (channel-first) field: the innermost batch dim runs along memory
            soa = keep_soa and nb > 0 and t.dim() > nb and t.stride(nb - 1) == 1 and t.stride(nb) != 1
            if bcast or soa or t.is_contiguous() or t.numel() == 0:
                new.append(t)
                continue
            c = t.contiguous()
            if k == len(tensors) - 1 and self._copyback is None:
                self._copyback = (t, c)
            new.append(c)
        return new

    def _materialise(self, tensors):
        new = [t.contiguous() for t in tensors]
        if new[-1] is not tensors[-1] and self._copyback is None:
            self._copyback = (tensors[-1], new[-1])
        return new

    def finish(self):
        """Copy a temporary output back into the user's strided `out=` tensor."""
        if self._copyback is not None:
            dst, src = self._copyback
            dst.copy_(src)


def expand_batch(batch_shape, t, ncomp):
    """Expand `t` (batch dims + ncomp component dims) to the broadcast batch shape (a view)."""
    nb = t.dim() - ncomp
    if t.shape[:nb] == tuple(batch_shape):       # nothing to broadcast: the common case, and `expand` is 1.5 us
        return t
    comp = tuple(t.shape[nb:]) if ncomp else ()
    return t.expand(tuple(batch_shape) + comp)
