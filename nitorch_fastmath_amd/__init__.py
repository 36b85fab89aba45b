"""nitorch_fastmath_amd -- MI355X-native (gfx950) backend for the per-element
small-matrix hot path of nitorch-fastmath: `sym`, `batched`, `qr` and the NaN-omitting
reductions of `reduce`, behind the reference's own Python function signatures.

Host code is Python on PyTorch-ROCm (device memory, streams); the arithmetic is
hand-written HIP in `libnfm_hip.so`, reached through the C ABI of include/nfm_hip.h.
There is no CPU path: importing works anywhere, calling needs the built library and
GPU tensors.
"""
from . import sym, batched, reduce, qr, utils  # noqa: F401
from .sym import *       # noqa: F401,F403
from .batched import *   # noqa: F401,F403
from .qr import *        # noqa: F401,F403
# `reduce` shadows builtins (min, max, sum) exactly like the reference's star-import does
# (`__init__.py:1-10`); keep them namespaced at package level as well.
from ._lib import LIB_PATH, lib as _load_lib  # noqa: F401


def is_built():
    """True when the HIP shared library has been compiled in-tree."""
    import os
    return os.path.exists(LIB_PATH)
