#!/usr/bin/env python
"""Gauss-Newton step: sym_solve(sym_matmul(J, H), g) chained vs the fused kernel."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402

dev = torch.device('cuda:0')


from _timing import timeit as _timeit

def timeit(fn, reps=10):
    return _timeit(fn, reps)           # seconds, steady state (scripts/_timing.py)



print('| k x d | dtype | n | chain ms | fused ms | speed-up | fused algorithmic GB/s |')
print('|---|---|---|---|---|---|---|')
g = torch.Generator(device=dev).manual_seed(0)
n = 30_000_000
for dt, dn, sz in ((torch.float32, 'f32', 4), (torch.float64, 'f64', 8)):
    for k, d in ((2, 2), (3, 3), (4, 4), (3, 2)):
        K = k * (k + 1) // 2
        j = torch.randn(n, k, d, device=dev, generator=g, dtype=dt)
        h = 0.2 * torch.randn(n, K, device=dev, generator=g, dtype=dt)
        h[:, :k] += 2
        b = torch.randn(n, d, device=dev, generator=g, dtype=dt)
        out = torch.empty_like(b)
        tc = timeit(lambda: N.sym_solve(N.sym_matmul(j, h), b, eps=1e-3, out=out))
        tf = timeit(lambda: N.sym.sym_matmul_solve(j, h, b, eps=1e-3, out=out))
        bpu = (k * d + K + 2 * d) * sz
        print(f'| {k}x{d} | {dn} | {n:.0e} | {tc * 1e3:.3f} | {tf * 1e3:.3f} | {tc / tf:.2f} | {n * bpu / tf / 1e9:.0f} |')
        del j, h, b, out
