#!/usr/bin/env python
"""cProfile of the facade's host side for small batches (launch-bound calls)."""
import cProfile
import os
import pstats
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
n = 1000
mat3 = torch.randn(n, 6, device=dev, generator=g, dtype=torch.float64)
mat3[:, :3] += 4
mat4 = torch.randn(n, 10, device=dev, generator=g)
mat4[:, :4] += 4
vec4 = torch.randn(n, 4, device=dev, generator=g)
for name, fn in (('sym_invert', lambda: N.sym_invert(mat3)), ('sym_solve', lambda: N.sym_solve(mat4, vec4))):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3000):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    print('=====', name)
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(14)
