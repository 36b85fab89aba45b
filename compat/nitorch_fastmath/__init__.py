"""Import-compatible facade: put `compat/` on sys.path and `import nitorch_fastmath` resolves
the hot-path modules (`sym`, `batched`, `qr`, `reduce`, and the helpers of `utils`) to the MI355X backend
`nitorch_fastmath_amd`.  Only the modules on the accelerated path exist here; the rest of
the upstream package (lie, realtransforms, simplex, special, stochastic, sugar) is out of
scope of this backend."""
from nitorch_fastmath_amd import sym, batched, qr, reduce, utils  # noqa: F401
from nitorch_fastmath_amd.sym import *       # noqa: F401,F403
from nitorch_fastmath_amd.batched import *   # noqa: F401,F403
from nitorch_fastmath_amd.qr import *        # noqa: F401,F403
from nitorch_fastmath_amd.reduce import *    # noqa: F401,F403
import sys as _sys
for _m in ('sym', 'batched', 'qr', 'reduce', 'utils'):
    _sys.modules[__name__ + '.' + _m] = globals()[_m]
