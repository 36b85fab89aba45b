import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N
dev = torch.device('cuda:0')
def timeit(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for M in (9, 12, 16):
    K = M * (M + 1) // 2
    n = 1 << 20
    g = torch.Generator(device=dev).manual_seed(M)
    mat = (0.3 * torch.randn(K, n, device=dev, generator=g) / M)
    mat[:M] += 2
    vec = torch.randn(M, n, device=dev, generator=g)
    mT, vT = mat.T, vec.T            # channel-first (SoA) views
    mc, vc = mT.contiguous(), vT.contiguous()
    t_soa = timeit(lambda: N.sym_solve(mT, vT))
    t_aos = timeit(lambda: N.sym_solve(mc, vc))
    r = (N.sym_solve(mT, vT) - N.sym_solve(mc, vc)).abs().max().item()
    ti_soa = timeit(lambda: N.sym_invert(mT)); ti_aos = timeit(lambda: N.sym_invert(mc))
    a = torch.randn(n // 4, M, M, device=dev, generator=g) + 6 * torch.eye(M, device=dev)
    aT = a.permute(1, 2, 0).contiguous().permute(2, 0, 1)
    tb_soa = timeit(lambda: N.batchinv(aT)); tb_aos = timeit(lambda: N.batchinv(a))
    print(f'M={M}: solve soa {t_soa:.3f} ms aos {t_aos:.3f} ms diff {r:.1e} | invert soa {ti_soa:.3f} aos {ti_aos:.3f} | batchinv soa {tb_soa:.3f} aos {tb_aos:.3f}')
