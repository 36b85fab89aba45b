#!/usr/bin/env python
"""qr family at orders 9..16 (run-time-order kernels)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402

dev = torch.device('cuda:0')


from _timing import timeit as _timeit

def timeit(fn, reps=3):
    return _timeit(fn, reps)           # seconds, steady state (scripts/_timing.py)



print('| op | n | ms | matrices/s | GB/s | frac of 8 TB/s |')
print('|---|---|---|---|---|---|')
g = torch.Generator(device=dev).manual_seed(0)
for dt, dn, sz in ((torch.float32, 'f32', 4), (torch.float64, 'f64', 8)):
    for M in (8, 9, 12, 16):
        n = 1 << 19
        A = torch.randn(n, M, M, device=dev, generator=g, dtype=dt)
        S = (A + A.transpose(-1, -2)).contiguous()
        H = torch.triu(A, -1)
        for name, fn, b in (('hessenberg', lambda: N.hessenberg(A, check_finite=False), 2 * M * M * sz),
                            ('hessenberg_sym', lambda: N.hessenberg_sym(S, check_finite=False), 2 * M * M * sz),
                            ('qr_hessenberg', lambda: N.qr_hessenberg(H, check_finite=False), 3 * M * M * sz),
                            ('rq_hessenberg', lambda: N.rq_hessenberg(H, check_finite=False), 2 * M * M * sz),
                            ('householder', lambda: N.householder(A[:, 0], check_finite=False), (2 * M + 1) * sz),
                            ('eig_sym values', lambda: N.eig_sym(S, check_finite=False), (M * M + M) * sz),
                            ('eig_sym vectors', lambda: N.eig_sym(S, compute_u=True, check_finite=False), (2 * M * M + M) * sz),
                            ("eig_sym values, arithmetic='fast'", lambda: N.eig_sym(S, check_finite=False, arithmetic='fast'), (M * M + M) * sz),
                            ("eig_sym vectors, arithmetic='fast'", lambda: N.eig_sym(S, compute_u=True, check_finite=False, arithmetic='fast'), (2 * M * M + M) * sz)):
            t = timeit(fn)
            print(f'| {name} {M}x{M} {dn} | {n} | {t * 1e3:.3f} | {n / t:.3e} | {n * b / t / 1e9:.0f} | {n * b / t / 8e12:.3f} |')
