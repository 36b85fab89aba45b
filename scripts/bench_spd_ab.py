#!/usr/bin/env python
"""no-exchange-first kernels (nfm_spd.hip) against the pivoted kernels they stand in front of, same process
arguments, one run per arm (NFM_DEBUG=1 NFM_SPD_OFF=1 is the pivoted arm): ms per call at several batch sizes."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nitorch_fastmath_amd as N  # noqa: E402
from _timing import timeit  # noqa: E402

dev = torch.device('cuda:0')
arm = 'pivoted' if os.environ.get('NFM_SPD_OFF') else 'no-exchange first'
print(f'# arm: {arm}')
print('| op | batch | ms | GB/s | frac of 8 TB/s |')
print('|---|---|---|---|---|')
# the compact symmetric functions: positive definite input (every matrix takes the unpivoted path), and the worst
# case -- a batch in which EVERY matrix is indefinite (the attempt is wasted and every group is redone)
for dtype, dn, sz in ((torch.float32, 'f32', 4), (torch.float64, 'f64', 8)):
    for M in (12, 16):
        K = M * (M + 1) // 2
        n = int(2e6)
        g = torch.Generator(device=dev).manual_seed(M)
        for what in ('positive definite', 'all indefinite'):
            mat = 0.3 * torch.randn(n, K, device=dev, generator=g, dtype=dtype) / M
            mat[:, :M] += 2
            if what == 'all indefinite':
                mat[:, 0] = -2
            vec = torch.randn(n, M, device=dev, generator=g, dtype=dtype)
            out = torch.empty_like(vec)
            t = timeit(lambda: N.sym_solve(mat, vec, out=out), 6)
            b = (K + 2 * M) * sz
            print(f'| sym_solve {M}x{M} {dn}, {what} | {n:.1e} | {t * 1e3:.3f} | {n * b / t / 1e9:.0f} | {n * b / t / 8e12:.3f} |')
            inv = torch.empty_like(mat)
            t = timeit(lambda: N.sym_invert(mat, out=inv), 6)
            b = 2 * K * sz
            print(f'| sym_invert {M}x{M} {dn}, {what} | {n:.1e} | {t * 1e3:.3f} | {n * b / t / 1e9:.0f} | {n * b / t / 8e12:.3f} |')
            del mat, vec, out, inv
for dtype, dn, sz in ((torch.float32, 'f32', 4), (torch.float64, 'f64', 8)):
    for Nn in (9, 12, 16):
        if dn == 'f64' and Nn > 11:
            continue
        for n in (int(6e5), int(2.4e6), int(4.8e6)):
            if n * Nn * Nn * sz * 2 > 40e9:
                continue
            g = torch.Generator(device=dev).manual_seed(Nn)
            a = torch.randn(n, Nn, Nn, device=dev, generator=g, dtype=dtype) + 8 * torch.eye(Nn, device=dev, dtype=dtype)
            t = timeit(lambda: N.batchdet(a), 6)
            b = (Nn * Nn + 1) * sz
            print(f'| batchdet {Nn}x{Nn} {dn} | {n:.1e} | {t * 1e3:.3f} | {n * b / t / 1e9:.0f} | {n * b / t / 8e12:.3f} |')
            t = timeit(lambda: N.batchinv(a), 6)
            b = 2 * Nn * Nn * sz
            print(f'| batchinv {Nn}x{Nn} {dn} | {n:.1e} | {t * 1e3:.3f} | {n * b / t / 1e9:.0f} | {n * b / t / 8e12:.3f} |')
            del a
