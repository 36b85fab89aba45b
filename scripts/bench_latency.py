#!/usr/bin/env python
"""Host-side cost per call of the facade (small batches: launch-bound)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)


def per_call(fn, reps=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6


n = 1000
mat3 = torch.randn(n, 6, device=dev, generator=g, dtype=torch.float64)
mat3[:, :3] += 4
mat4 = torch.randn(n, 10, device=dev, generator=g)
mat4[:, :4] += 4
vec4 = torch.randn(n, 4, device=dev, generator=g)
out4 = torch.empty_like(vec4)
a8 = torch.randn(n, 8, 8, device=dev, generator=g, dtype=torch.float64) + 8 * torch.eye(8, device=dev, dtype=torch.float64)
x = torch.randn(n, 64, device=dev, generator=g)
rows = [
    ('torch.add (reference point)', lambda: torch.add(vec4, vec4)),
    ('sym_invert 3x3 f64', lambda: N.sym_invert(mat3)),
    ('sym_solve 4x4 f32', lambda: N.sym_solve(mat4, vec4)),
    ('sym_solve 4x4 f32 out=', lambda: N.sym_solve(mat4, vec4, out=out4)),
    ('sym_matvec 4x4 f32', lambda: N.sym_matvec(mat4, vec4)),
    ('batchinv 8x8 f64', lambda: N.batchinv(a8)),
    ('nansum dim=None', lambda: N.reduce.nansum(x)),
    ('nansum dim=1', lambda: N.reduce.nansum(x, dim=1)),
    ('nanmax dim=1 +idx', lambda: N.reduce.nanmax(x, dim=1, return_indices=True)),
    ('nanvar dim=1', lambda: N.reduce.nanvar(x, dim=1)),
    ('eig_sym 3x3', lambda: N.eig_sym(a8[:, :3, :3] + a8[:, :3, :3].transpose(-1, -2), check_finite=False)),
]
print('| call (batch 1000) | host us/call (enqueue) | us/call incl. drain |')
print('|---|---|---|')
for name, fn in rows:
    a, b = per_call(fn)
    print(f'| {name} | {a:.1f} | {b:.1f} |')
