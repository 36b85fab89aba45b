#!/usr/bin/env python
"""Throughput of the dim-wise reductions and of the qr family (hipEvent timing through the facade).
Prints markdown: algorithmic GB/s (input bytes + output bytes) and fraction of the 8 TB/s roofline."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402
from nitorch_fastmath_amd import reduce as R  # noqa: E402

dev = torch.device('cuda:0')


from _timing import timeit as _timeit

def timeit(fn, reps=8):
    return _timeit(fn, reps)           # seconds, steady state (scripts/_timing.py)



rows = []
which = os.environ.get('WHICH', 'reduce,qr').split(',')
if 'reduce' in which:
    total = 1 << int(os.environ.get('LOG2N', '30'))
    x = torch.randn(total, device=dev)
    x[::97] = float('nan')
    shapes = [  # (outer, red, inner)
        (total // 8, 8, 1), (total // 32, 32, 1), (total // 256, 256, 1), (total // 4096, 4096, 1),
        (total >> 16, 1 << 16, 1), (16, total // 16, 1),
        (1, 8, total // 8), (1, 256, total // 256), (64, 64, total // 4096), (1, 1 << 16, total >> 16),
        (total >> 12, 64, 64), (total >> 10, 256, 4), (1, total >> 4, 16),
    ]
    for (o, r, i) in shapes:
        v = x.view(o, r, i)
        nout = o * i
        for name, fn in (('nansum', lambda: R.nansum(v, dim=1)),
                         ('nanmax', lambda: R.nanmax(v, dim=1)),
                         ('nanmax+idx', lambda: R.nanmax(v, dim=1, return_indices=True)),
                         ('nanvar', lambda: R.nanvar(v, dim=1))):
            t = timeit(fn, reps=5)
            b = total * 4 + nout * (12 if name.endswith('idx') else 4)
            rows.append((f'{name} ({o},{r},{i}) dim=1', total, b / total, t))
    del x
if 'qr' in which:
    for dtype, dn, sz in ((torch.float32, 'f32', 4), (torch.float64, 'f64', 8)):
        n = 1 << 24
        g = torch.Generator(device=dev).manual_seed(1)
        a = torch.randn(n, device=dev, generator=g, dtype=dtype)
        b = torch.randn(n, device=dev, generator=g, dtype=dtype)
        rows.append((f'givens {dn}', n, 4 * sz, timeit(lambda: N.givens(a, b))))
        for M in (3, 4, 6, 8):
            n = int(min(2e7, 1.5e9 / (3 * M * M * sz)))
            A = torch.randn(n, M, M, device=dev, generator=g, dtype=dtype)
            S = A + A.transpose(-1, -2)
            v = torch.randn(n, M, device=dev, generator=g, dtype=dtype)
            rows.append((f'householder {M} {dn}', n, (2 * M + 1) * sz, timeit(lambda: N.householder(v, check_finite=False))))
            rows.append((f'hessenberg {M}x{M} {dn}', n, 2 * M * M * sz, timeit(lambda: N.hessenberg(A, check_finite=False))))
            rows.append((f'hessenberg_sym {M}x{M} {dn}', n, 2 * M * M * sz, timeit(lambda: N.hessenberg_sym(S, check_finite=False))))
            H = torch.triu(A, -1)
            rows.append((f'qr_hessenberg {M}x{M} {dn}', n, 3 * M * M * sz, timeit(lambda: N.qr_hessenberg(H, check_finite=False))))
            rows.append((f'rq_hessenberg {M}x{M} {dn}', n, 2 * M * M * sz, timeit(lambda: N.rq_hessenberg(H, check_finite=False))))
            for mode in ('reference', 'fast'):      # 'reference' is the default (bit-identical to the CPU path)
                lab = '' if mode == 'reference' else ", arithmetic='fast'"
                rows.append((f'eig_sym {M}x{M} {dn} values{lab}', n, (M * M + M) * sz,
                             timeit(lambda: N.eig_sym(S, check_finite=False, arithmetic=mode), reps=3)))
                rows.append((f'eig_sym {M}x{M} {dn} vectors{lab}', n, (2 * M * M + M) * sz,
                             timeit(lambda: N.eig_sym(S, compute_u=True, check_finite=False, arithmetic=mode), reps=3)))
            del A, S, v, H
print('| op | units | B/unit | ms | units/s | GB/s | frac of 8 TB/s |')
print('|---|---|---|---|---|---|---|')
for name, n, b, t in rows:
    print(f'| {name} | {n:.2e} | {b:.1f} | {t * 1e3:.3f} | {n / t:.3e} | {n * b / t / 1e9:.0f} | {n * b / t / 8e12:.3f} |')
