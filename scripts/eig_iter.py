import os, sys, torch
sys.path.insert(0, os.getcwd())
import nitorch_fastmath_amd as N
dev = torch.device('cuda:0')
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
g = torch.Generator(device=dev).manual_seed(0)
for dt in (torch.float32, torch.float64):
    for M in (3, 4, 6):
        n = 1 << 21
        A = torch.randn(n, M, M, device=dev, generator=g, dtype=dt)
        S = (A + A.transpose(-1, -2)).contiguous()
        ref = torch.linalg.eigvalsh(S[:4096].double().cpu())
        for mi in (8, 16, 32, 64, 1024):
            t = timeit(lambda: N.eig_sym(S, check_finite=False, max_iter=mi))
            v = N.eig_sym(S[:4096], check_finite=False, max_iter=mi).double().cpu().sort(-1).values
            err = ((v - ref).abs().max() / ref.abs().max()).item()
            tu = timeit(lambda: N.eig_sym(S, compute_u=True, check_finite=False, max_iter=mi))
            print(f'{dt} M={M} max_iter={mi}: values {t:.3f} ms ({n/t/1e6:.2f} G/s) err {err:.1e} | vectors {tu:.3f} ms')
