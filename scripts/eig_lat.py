import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import nitorch_fastmath_amd as N
dev = torch.device('cuda:0')
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
g = torch.Generator(device=dev).manual_seed(0)
for n in (1000, 100000):
    a8 = torch.randn(n, 8, 8, device=dev, generator=g, dtype=torch.float64) + 8 * torch.eye(8, device=dev, dtype=torch.float64)
    S = (a8[:, :3, :3] + a8[:, :3, :3].transpose(-1, -2)).contiguous()
    for mi in (4, 8, 16, 64, 1024):
        t = timeit(lambda: N.eig_sym(S, check_finite=False, max_iter=mi))
        print(f'n={n} f64 shifted max_iter={mi}: {t*1e3:.1f} us')
    S32 = S.float()
    for mi in (8, 1024):
        t = timeit(lambda: N.eig_sym(S32, check_finite=False, max_iter=mi))
        print(f'n={n} f32 shifted max_iter={mi}: {t*1e3:.1f} us')
    t0 = time.perf_counter()
    for _ in range(200):
        N.eig_sym(S, check_finite=False, max_iter=8)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f'host per call {1e6*(t1-t0)/200:.1f} us')
