#!/usr/bin/env python
"""Small-record ops (4 / 8 / 16-byte records, one packed access per lane) at batch sizes from the
qr table's 1.7e7 up to 2e8: is their 43-60 % of the roofline a size effect or the access width?
Prints a markdown table (+ a torch element-wise op of the same byte count for reference)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402

dev = torch.device('cuda:0')


from _timing import timeit as _timeit

def timeit(fn, reps=10):
    return _timeit(fn, reps)           # seconds, steady state (scripts/_timing.py)



print('| op | units | B/unit | ms | GB/s | frac of 8 TB/s |')
print('|---|---|---|---|---|---|')
g = torch.Generator(device=dev).manual_seed(0)
for n in (1.68e7, 6e7, 2e8):
    n = int(n)
    x, y = torch.randn(n, device=dev, generator=g), torch.randn(n, device=dev, generator=g)
    rows = [('givens f32', lambda: N.givens(x, y), 16),
            ('torch x*y+x (same 12 B/unit)', lambda: torch.addcmul(x, x, y), 12)]
    v3 = torch.randn(n // 2, 3, device=dev, generator=g)
    rows.append(('householder 3 f32', lambda: N.householder(v3, check_finite=False), 28 / 2))
    m1 = torch.rand(n, 1, device=dev, generator=g) + 1
    v1 = torch.randn(n, 1, device=dev, generator=g)
    rows.append(('sym_solve 1x1 f32', lambda: N.sym_solve(m1, v1), 12))
    a2 = torch.randn(n // 4, 2, 2, device=dev, generator=g) + 3 * torch.eye(2, device=dev)
    rows.append(('batchdet 2x2 f32', lambda: N.batchdet(a2), 20 / 4))
    rows.append(('batchinv 2x2 f32', lambda: N.batchinv(a2), 32 / 4))
    for name, fn, b in rows:
        t = timeit(fn)
        print(f'| {name} | {n:.2e} | {b:.1f} | {t * 1e3:.3f} | {n * b / t / 1e9:.0f} | {n * b / t / 8e12:.3f} |')
    del x, y, v3, m1, v1, a2
