import os, sys, torch
sys.path.insert(0, os.getcwd())
import nitorch_fastmath_amd as N
dev = torch.device('cuda:0')
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best
for shape in ((2, 27, 192, 192, 192), (2, 27, 191, 193, 189), (1, 27, 14155777), (54, 27, 262144), (54, 27, 262147), (8, 8, 160, 160, 160), (4100, 27, 37), (100000, 8, 3)):
    x = torch.randn(shape, device=dev)
    t = timeit(lambda: N.reduce.median(x, dim=1))
    ti = timeit(lambda: N.reduce.median(x, dim=1, return_indices=True))
    print(shape, f'{t*1e3:.3f} ms {x.numel()*4/t/1e9:.0f} GB/s; with indices {ti*1e3:.3f} ms')
    del x
