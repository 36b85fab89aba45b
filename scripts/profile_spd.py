#!/usr/bin/env python
"""The no-exchange-first kernels of orders 9..16 under `rocprofv3 --kernel-trace --stats`: the throughput table's rows
at 12x12 / 16x16 (same batch sizes, positive definite / diagonally dominant input), 60 calls each.  The kernel
durations of the trace are the cross-check of the table's event timing (which goes through the facade)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402

dev = torch.device('cuda:0')
for dtype, sz in ((torch.float32, 4), (torch.float64, 8)):
    for M in (12, 16):
        K = M * (M + 1) // 2
        n = int(min(4e7, 2.5e9 / ((K + 2 * M) * sz))) // 2
        g = torch.Generator(device=dev).manual_seed(M)
        mat = 0.3 * torch.randn(n, K, device=dev, generator=g, dtype=dtype) / M
        mat[:, :M] += 2
        vec = torch.randn(n, M, device=dev, generator=g, dtype=dtype)
        out, inv = torch.empty_like(vec), torch.empty_like(mat)
        for _ in range(60):
            N.sym_solve(mat, vec, out=out)
            N.sym_invert(mat, out=inv)
        del mat, vec, out, inv
        if dtype == torch.float64:
            continue
        n = int(min(4e7, 2.5e9 / (2 * M * M * sz))) // 2
        a = torch.randn(n, M, M, device=dev, generator=g, dtype=dtype) + 6 * torch.eye(M, device=dev, dtype=dtype)
        for _ in range(60):
            N.batchinv(a)
            N.batchdet(a)
        del a
torch.cuda.synchronize()
