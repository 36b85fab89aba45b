#!/usr/bin/env python
"""Launch one row-wave case repeatedly (for rocprofv3 counter passes).
usage: rowwave_probe.py <op> <order> <f32|f64> [n]   op in sym_solve|sym_invert|sym_det|batchinv|batchdet"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402

op, M, dn = sys.argv[1], int(sys.argv[2]), sys.argv[3]
n = int(float(sys.argv[4])) if len(sys.argv) > 4 else 300_000
dev = torch.device('cuda:0')
dtype = torch.float32 if dn == 'f32' else torch.float64
g = torch.Generator(device=dev).manual_seed(M)
K = M * (M + 1) // 2
mat = 0.3 * torch.randn(n, K, device=dev, generator=g, dtype=dtype) / M
mat[:, :M] += 2
vec = torch.randn(n, M, device=dev, generator=g, dtype=dtype)
a = torch.randn(n, M, M, device=dev, generator=g, dtype=dtype) + 6 * torch.eye(M, device=dev, dtype=dtype)
fn = {'sym_solve': lambda: N.sym_solve(mat, vec), 'sym_invert': lambda: N.sym_invert(mat), 'sym_det': lambda: N.sym_det(mat),
      'batchinv': lambda: N.batchinv(a), 'batchdet': lambda: N.batchdet(a)}[op]
for _ in range(6):
    fn()
torch.cuda.synchronize()
