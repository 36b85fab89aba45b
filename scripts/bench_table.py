#!/usr/bin/env python
"""Throughput table over orders / dtypes / ops (hipEvent timing through the facade).
Prints markdown: units/s, algorithmic GB/s, fraction of the 8 TB/s roofline."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402

dev = torch.device('cuda:0')


from _timing import timeit as _timeit

def timeit(fn, reps=8):
    return _timeit(fn, reps)           # seconds, steady state (scripts/_timing.py)



def spd(n, M, dtype):
    g = torch.Generator(device=dev).manual_seed(M)
    K = M * (M + 1) // 2
    mat = 0.3 * torch.randn(n, K, device=dev, generator=g, dtype=dtype) / M
    mat[:, :M] += 2
    return mat, torch.randn(n, M, device=dev, generator=g, dtype=dtype)


rows = []
for dtype, dn, sz in ((torch.float32, 'f32', 4), (torch.float64, 'f64', 8)):
    for M in [int(v) for v in os.environ.get('ORDERS_SYM', '2,3,4,5,6,8,9,12,16').split(',')]:
        K = M * (M + 1) // 2
        n = int(min(4e7, 2.5e9 / ((K + 2 * M) * sz)))
        if M > 8:
            n = n // 2
        mat, vec = spd(n, M, dtype)
        out = torch.empty_like(vec)
        t = timeit(lambda: N.sym_solve(mat, vec, out=out))
        rows.append((f'sym_solve {M}x{M} {dn}', n, (K + 2 * M) * sz, t))
        inv = torch.empty_like(mat)
        t = timeit(lambda: N.sym_invert(mat, out=inv))
        rows.append((f'sym_invert {M}x{M} {dn}', n, 2 * K * sz, t))
        t = timeit(lambda: N.sym_matvec(mat, vec, out=out))
        rows.append((f'sym_matvec {M}x{M} {dn}', n, (K + 2 * M) * sz, t))
        del mat, vec, out, inv
    for Nn in [int(v) for v in os.environ.get('ORDERS_GEN', '2,3,4,6,8,12,16').split(',')]:
        n = int(min(4e7, 2.5e9 / (2 * Nn * Nn * sz)))
        if Nn > 8:
            n = n // 2
        g = torch.Generator(device=dev).manual_seed(Nn)
        a = torch.randn(n, Nn, Nn, device=dev, generator=g, dtype=dtype) + 6 * torch.eye(Nn, device=dev, dtype=dtype)
        t = timeit(lambda: N.batchinv(a))
        rows.append((f'batchinv {Nn}x{Nn} {dn}', n, 2 * Nn * Nn * sz, t))
        t = timeit(lambda: N.batchdet(a))
        rows.append((f'batchdet {Nn}x{Nn} {dn}', n, (Nn * Nn + 1) * sz, t))
        if Nn <= 8:
            s = a + a.transpose(-1, -2)
            t = timeit(lambda: N.eig_sym(s, check_finite=False), reps=4)
            rows.append((f'eig_sym {Nn}x{Nn} {dn} (values)', n, (Nn * Nn + Nn) * sz, t))
        del a
print('| op | batch | B/unit | ms | units/s | GB/s | frac of 8 TB/s |')
print('|---|---|---|---|---|---|---|')
for name, n, b, t in rows:
    print(f'| {name} | {n:.2e} | {b} | {t * 1e3:.3f} | {n / t:.3e} | {n * b / t / 1e9:.0f} | {n * b / t / 8e12:.3f} |')
