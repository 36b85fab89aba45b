"""Helpers reached from the hot path (reference `utils.py`): list coercion, linear ->
sub indices, machine epsilon."""
from types import GeneratorType as generator
import torch

__all__ = ['ensure_list', 'ind2sub', 'sub2ind', 'eps']


def ensure_list(x, size=None, crop=True, **kwargs):
    """Ensure that an object is a list (of size at least `size`) -- `utils.py:11-28`."""
    if not isinstance(x, (list, tuple, range, generator)):
        x = [x]
    elif not isinstance(x, list):
        x = list(x)
    if size and len(x) < size:
        default = kwargs.get('default', x[-1])
        x += [default] * (size - len(x))
    if size and crop:
        x = x[:size]
    return x


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


def sub2ind(subs, shape, out=None):
    """Sub indices `(D, ...)` -> linear indices (`utils.py:148-178`)."""
    subs = [torch.as_tensor(s) for s in subs]
    ind = torch.zeros_like(subs[-1]) if out is None else out.zero_()
    for s, st in zip(subs, _strides(shape)):
        ind += s * st
    return ind


def eps(dtype='float32'):
    """Machine epsilon table of the reference (`utils.py:232-249`)."""
    if dtype in ('float16', torch.float16, 'complex32', getattr(torch, 'complex32', None)):
        return 2 ** -10
    if dtype in ('float32', torch.float32, 'complex64', torch.complex64):
        return 2 ** -23
    if dtype in ('float64', torch.float64, 'complex128', torch.complex128):
        return 2 ** -52
    raise NotImplementedError
