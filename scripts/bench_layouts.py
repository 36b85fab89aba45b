#!/usr/bin/env python
"""4x4 / 6x6 fp32 sym_solve under the operand layouts the facade accepts without copies."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402

dev = torch.device('cuda:0')


from _timing import timeit as _timeit

def timeit(fn, reps=8):
    return _timeit(fn, reps)           # seconds, steady state (scripts/_timing.py)



print('| layout | M | n | ms | solves/s | algorithmic GB/s |')
print('|---|---|---|---|---|---|')
for M in (4, 6, 12, 16):
    K = M * (M + 1) // 2
    n = 17_000_000 if M <= 8 else 4_250_000   # not a power of two: component planes 2^k bytes apart camp on one HBM channel
    g = torch.Generator(device=dev).manual_seed(M)
    mat = 0.3 * torch.randn(n, K, device=dev, generator=g) / M
    mat[:, :M] += 2
    vec = torch.randn(n, M, device=dev, generator=g)
    out = torch.empty_like(vec)
    cases = []
    cases.append(('contiguous AoS', mat, vec, (K + 2 * M) * 4))
    cases.append(('channel-first (SoA) views', mat.T.contiguous().T, vec.T.contiguous().T, (K + 2 * M) * 4))
    nodd = n - 1
    mo = torch.empty(K, nodd, device=dev).copy_(mat[:nodd].T).T
    vo = torch.empty(M, nodd, device=dev).copy_(vec[:nodd].T).T
    cases.append(('channel-first, odd voxel count (runs start at any alignment)', mo, vo, (K + 2 * M) * 4))
    cases.append(('one matrix, n vectors (broadcast mat)', mat[:1], vec, 2 * M * 4))
    if M > 8:    # the same at the batch of the small orders (the matrix is factored once per workgroup: a fixed cost)
        cases.append(('one matrix, n vectors (broadcast mat), 1.7e7 vectors', mat[:1],
                      torch.randn(17_000_000, M, device=dev, generator=g), 2 * M * 4))
    cases.append(('n matrices, one vector (broadcast vec)', mat, vec[:1], (K + M) * 4))
    cases.append(('every other record (batch stride 2)', mat[::2], vec[::2], (K + 2 * M) * 4))
    big = torch.randn(n // 2, K, 2, device=dev, generator=g)
    big[:, :M, 0] += 4
    cases.append(('component stride 2', big[..., 0], vec[:n // 2], (K + 2 * M) * 4))
    m3 = mat.view(n // 4250, 4250, K)[:, ::2]
    v3 = vec.view(n // 4250, 4250, M)[:, ::2]
    cases.append(('two-level batch (rows of a strided 2-D field)', m3, v3, (K + 2 * M) * 4))
    pad = torch.zeros(n // 2, K + 2, device=dev)
    pad[:, :K] = mat[:n // 2]
    cases.append(('padded records: (n, K + 2)[:, :K]', pad[:, :K], vec[:n // 2], (K + 2 * M) * 4))
    mis = torch.empty(n * K + 1, device=dev)[1:].view(n, K).copy_(mat)
    cases.append(('base pointer off by 4 bytes', mis, vec, (K + 2 * M) * 4))
    cases.append(('rows 1.. of the contiguous tensors (x[1:])', mat[1:], vec[1:], (K + 2 * M) * 4))
    for name, m, v, bpu in cases:
        nn = max(m.shape[:-1].numel(), v.shape[:-1].numel())
        t = timeit(lambda: N.sym_solve(m, v))
        print(f'| {name} | {M} | {nn:.2e} | {t * 1e3:.3f} | {nn / t:.3e} | {nn * bpu / t / 1e9:.0f} |')
