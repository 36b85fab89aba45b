import os, sys, cProfile, pstats, torch
sys.path.insert(0, os.getcwd())
import nitorch_fastmath_amd as N
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
n = 1000
mat4 = torch.randn(n, 10, device=dev, generator=g); mat4[:, :4] += 4
vec4 = torch.randn(n, 4, device=dev, generator=g)
for _ in range(100): N.sym_solve(mat4, vec4)
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): N.sym_solve(mat4, vec4)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
