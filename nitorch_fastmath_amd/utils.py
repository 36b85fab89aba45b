"""Helpers reached from the hot path (`ensure_list`, `ind2sub`, `eps` of the
reference's `utils.py`; its other helpers are not on the path and are not rebuilt), plus
`graphed` (HIP-graph capture of launch-bound call chains)."""
from types import GeneratorType as generator
import torch

__all__ = ['ensure_list', 'ind2sub', 'eps', 'graphed']


def ensure_list(x, size=None, crop=True, **kwargs):
    """Coerce `x` to a list; with `size`, pad it to at least `size` items (with `default=` or,
    by default, its last item) and, unless `crop=False`, cut it to exactly `size`.
    Same contract as the reference helper (`utils.py:11-28`): scalars and other non-sequences
    become one-item lists; lists are padded in place, tuples / ranges / generators are copied."""
    if isinstance(x, list):
        items = x
    elif isinstance(x, (tuple, range, generator)):
        items = list(x)
    else:
        items = [x]
    if size:
        missing = size - len(items)
        if missing > 0:
            fill = kwargs['default'] if 'default' in kwargs else items[-1]
            items.extend([fill] * missing)
        if crop:
            items = items[:size]
    return items


def _strides(shape):
    out, acc = [], 1
    for s in reversed(list(shape)):
        out.append(acc)
        acc *= int(s)
    return list(reversed(out))


def ind2sub(ind, shape, out=None):
    """Linear indices -> sub indices `(D, ...)`, rightmost dimension fastest (`utils.py:196-229`)."""
    ind = torch.as_tensor(ind)
    sub = ind.new_empty([len(shape), *ind.shape]) if out is None else out.reshape([len(shape), *ind.shape])
    rem = ind
    for d, st in enumerate(_strides(shape)):
        sub[d] = torch.div(rem, st, rounding_mode='trunc')
        rem = rem - sub[d] * st
    return sub


def eps(dtype='float32'):
    """Machine epsilon table of the reference (`utils.py:232-249`)."""
    if dtype in ('float16', torch.float16, 'complex32', getattr(torch, 'complex32', None)):
        return 2 ** -10
    if dtype in ('float32', torch.float32, 'complex64', torch.complex64):
        return 2 ** -23
    if dtype in ('float64', torch.float64, 'complex128', torch.complex128):
        return 2 ** -52
    raise NotImplementedError


class graphed:
    """Capture a launch-bound sequence of calls into one HIP graph and replay it.

    Small batches (the reference's 1e5-matrix configuration, Gauss-Newton inner loops over a
    few thousand voxels) spend 15-30 us per call on the host for kernels of a few us; a graph
    replays the whole captured sequence with one launch.  The C entry points never allocate or
    synchronise, so any composition of them captures -- with two facade-level exceptions, both host
    reads that cannot be recorded into a graph: the `qr` functions must be called with
    `check_finite=False` (their default `check_finite=True` runs `torch.isfinite(a).all()` and
    reads the result on the host), and reductions must not be read back (`float(nansum(x))`)
    inside the captured function.

        step = graphed(lambda h, g: sym_solve(h, g), hess, grad)   # captures once (static buffers)
        x = step(hess_new, grad_new)                                # copies in, replays, returns outputs

    Arguments must keep their shapes / dtypes; the returned tensors are the graph's static
    outputs (overwritten by the next replay; clone to keep)."""

    def __init__(self, fn, *example_args, warmup=2):
        import torch
        self._torch = torch
        self._static_in = [a.clone() if isinstance(a, torch.Tensor) else a for a in example_args]
        dev = next(a.device for a in self._static_in if isinstance(a, torch.Tensor))
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():      # warm-up off the capture stream
            for _ in range(warmup):
                fn(*self._static_in)
        torch.cuda.current_stream(dev).wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph), torch.no_grad():
            self._static_out = fn(*self._static_in)

    def __call__(self, *args):
        torch = self._torch
        for dst, src in zip(self._static_in, args):
            if isinstance(dst, torch.Tensor) and src is not dst:
                dst.copy_(src, non_blocking=True)
        self._graph.replay()
        return self._static_out
