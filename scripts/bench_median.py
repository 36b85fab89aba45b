#!/usr/bin/env python
"""median (radix selection, nfm_reduce_median) against torch.median on the same device.
usage: bench_median.py [--no-torch] > profiles/rNN/median_table.md"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402

dev = torch.device('cuda:0')
with_torch = '--no-torch' not in sys.argv


from _timing import timeit as _timeit

def timeit(fn, reps=5):
    return _timeit(fn, reps)           # seconds, steady state (scripts/_timing.py)



print('# median: radix selection (nfm_reduce_median) vs torch.median on the same MI355X, contiguous (rows, red)\n')
print('| dtype | shape (rows x red) | ours ms | GB/s of one pass | torch.median ms |')
print('|---|---|---|---|---|')
for dtype, shapes in ((torch.float32, ((1, 1 << 30), (64, 1 << 24), (1 << 14, 1 << 16), (1 << 20, 1024), (1 << 22, 256),
                                       (1 << 22, 129), (1 << 22, 128), (1 << 22, 125), (1 << 23, 81), (1 << 23, 64), (1 << 23, 49), (1 << 24, 27), (1 << 25, 8))),
                      (torch.float64, ((1, 1 << 29), (1 << 13, 1 << 16), (1 << 19, 1024), (1 << 22, 65), (1 << 22, 64), (1 << 23, 27), (1 << 24, 8)))):
    for rows, red in shapes:
        x = torch.randn(rows, red, device=dev, dtype=dtype)
        t = timeit(lambda: N.reduce.median(x, dim=1))
        tt = float('nan')
        if with_torch:
            try:
                tt = timeit(lambda: torch.median(x, dim=1), reps=1)
            except Exception:
                pass
        print(f'| {str(dtype)[6:]} | {rows} x {red} | {t * 1e3:.3f} | {rows * red * x.element_size() / t / 1e9:.0f} | {tt * 1e3:.3f} |')
        del x
print('\nGB/s = rows x red x element size / time: one pass over the data.  Rows longer than 1024 make one streaming pass '
      'per 11-bit digit of the key (3 for float32, 6 for float64); rows up to 1024 are read once (keys stay in registers).  '
      'torch.median sorts every row (and is serial within one 2^30-element row).')

print('\n## the channel dim of a channel-first field: median(x, dim=1), x of shape (B, C, X, Y, Z)\n')
print('| dtype | shape | ours ms (no transposing copy) | GB/s | the same through a copy with the dim moved last, ms | torch.median ms |')
print('|---|---|---|---|---|---|')
for dtype, shape in ((torch.float32, (2, 27, 192, 192, 192)), (torch.float32, (8, 8, 160, 160, 160)),
                     (torch.float64, (2, 27, 160, 160, 160))):
    x = torch.randn(shape, device=dev, dtype=dtype)
    t = timeit(lambda: N.reduce.median(x, dim=1))
    tc = timeit(lambda: N.reduce.median(x.movedim(1, -1).contiguous(), dim=-1))
    tt = timeit(lambda: torch.median(x, dim=1), reps=1) if with_torch else float('nan')
    print(f'| {str(dtype)[6:]} | {tuple(shape)} | {t * 1e3:.3f} | {x.numel() * x.element_size() / t / 1e9:.0f} | {tc * 1e3:.3f} | {tt * 1e3:.3f} |')
    del x
