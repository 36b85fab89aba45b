#!/usr/bin/env python
"""Throughput of the remaining sym/batched entry points: algorithmic GB/s."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402

dev = torch.device('cuda:0')


from _timing import timeit as _timeit

def timeit(fn, reps=8):
    return _timeit(fn, reps)           # seconds, steady state (scripts/_timing.py)



rows = []
g = torch.Generator(device=dev).manual_seed(0)
n = 20_000_000
for dt, dn, sz in ((torch.float32, 'f32', 4), (torch.float64, 'f64', 8)):
    for M in (2, 3, 4, 6):
        K = M * (M + 1) // 2
        mat = torch.randn(n, K, device=dev, generator=g, dtype=dt)
        mat[:, :M] += 4
        vec = torch.randn(n, M, device=dev, generator=g, dtype=dt)
        rows.append((f'sym_det {M}x{M} {dn}', n, (K + 1) * sz, timeit(lambda: N.sym_det(mat))))
        rows.append((f'sym_to_full {M}x{M} {dn}', n, (K + M * M) * sz, timeit(lambda: N.sym_to_full(mat))))
        rows.append((f'sym_outer {M} {dn}', n, (M + K) * sz, timeit(lambda: N.sym_outer(vec))))
        rows.append((f'sym_invert diag {M}x{M} {dn}', n, (K + M) * sz, timeit(lambda: N.sym_invert(mat, diag=True))))
        rows.append((f'sym_addmatvec {M}x{M} {dn}', n, (K + 3 * M) * sz, timeit(lambda: N.sym_addmatvec(vec, mat, vec))))
        a = torch.randn(n, M, M, device=dev, generator=g, dtype=dt)
        rows.append((f'batchmatvec {M}x{M} {dn}', n, (M * M + 2 * M) * sz, timeit(lambda: N.batchmatvec(a, vec))))
        rows.append((f'batchdet {M}x{M} {dn}', n, (M * M + 1) * sz, timeit(lambda: N.batchdet(a))))
        for d in (2, 3):
            if M <= 4:
                j = torch.randn(n, M, d, device=dev, generator=g, dtype=dt)
                Kd = d * (d + 1) // 2
                rows.append((f'sym_matmul J({M}x{d}) H({M}x{M}) {dn}', n, (M * d + K + Kd) * sz,
                             timeit(lambda: N.sym_matmul(j, mat))))
                del j
        del mat, vec, a
print('| op | batch | B/unit | ms | units/s | GB/s | frac of 8 TB/s |')
print('|---|---|---|---|---|---|---|')
for name, nn, b, t in rows:
    print(f'| {name} | {nn:.1e} | {b} | {t * 1e3:.3f} | {nn / t:.3e} | {nn * b / t / 1e9:.0f} | {nn * b / t / 8e12:.3f} |')
