"""Shared timing helper of the table scripts: steady-state launch time by HIP events.

The first ~10 launches after an idle gap run up to 30 % slower (power-state ramp, bench.py's settle
phase exists for the same reason), so two warm-up calls are not enough for sub-millisecond kernels:
`fn` is launched back to back for `settle_ms` first, then `reps` launches are timed one by one and
the MEDIAN is returned (seconds) -- round 2's tables quoted the minimum after two warm-ups."""
import time
import torch


def timeit(fn, reps=8, settle_ms=40.0, stat='median'):
    fn()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + settle_ms * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(max(3, reps)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    return ts[0] if stat == 'min' else ts[len(ts) // 2]
