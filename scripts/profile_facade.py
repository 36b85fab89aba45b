"""cProfile of the facade's host path for launch-bound calls (where do the 12 us go?)"""
import cProfile, pstats, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N
dev = torch.device('cuda:0')
mat3 = torch.randn(1000, 6, device=dev, dtype=torch.float64); mat3[:, :3] += 4
mat4 = torch.randn(1000, 10, device=dev); mat4[:, :4] += 4
vec4 = torch.randn(1000, 4, device=dev)
for name, fn in (('sym_invert', lambda: N.sym_invert(mat3)), ('sym_solve', lambda: N.sym_solve(mat4, vec4))):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5000): fn()
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
    print('=====', name); print('\n'.join(s.getvalue().split('\n')[4:34]))
    t0 = time.perf_counter()
    for _ in range(5000): fn()
    print('us/call without profiler:', (time.perf_counter() - t0) / 5000 * 1e6)
