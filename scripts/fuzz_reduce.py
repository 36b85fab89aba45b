#!/usr/bin/env python
"""Random-shape fuzz of the dim-wise reductions against numpy's nan-functions.
usage: fuzz_reduce.py [seconds] [seed]"""
import os
import sys
import time
import warnings
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nitorch_fastmath_amd import reduce as R  # noqa: E402

dev = torch.device('cuda:0')
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
warnings.simplefilter('ignore')

SMALL = [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 17, 24, 31, 32, 33, 48, 63, 64, 65, 100, 128, 129, 255, 256, 257]
BIG = [300, 511, 512, 1000, 1024, 1025, 4096, 5000, 20_000, 65_536, 100_003, 300_000]


def pick_shape():
    cap = 3_000_000
    while True:
        o = int(rng.choice(SMALL + BIG + [1, 1, 1]))
        r = int(rng.choice(SMALL + BIG))
        i = int(rng.choice(SMALL + BIG + [1, 1, 1, 1]))
        if o * r * i <= cap:
            return o, r, i


def check(x, xd, what):
    x64 = x.astype(np.float64)
    f32 = x.dtype == np.float32
    tol = 2e-6 if f32 else 1e-12
    e = np.nansum(x64, axis=1)
    scale = np.nansum(np.abs(x64), axis=1) + 1e-300
    r = R.nansum(xd, dim=1, dtype=torch.float64).cpu().numpy()
    assert (np.abs(r - e) <= 1e-12 * scale).all(), (what, 'nansum64')
    assert (np.abs(R.nansum(xd, dim=1).cpu().numpy() - e) <= tol * scale).all(), (what, 'nansum')
    for name, fill, arg in (('nanmax', -np.inf, 'argmax'), ('nanmin', np.inf, 'argmin')):
        xx = np.where(np.isnan(x), fill, x)
        v, i = getattr(R, name)(xd, dim=1, return_indices=True)
        assert np.array_equal(v.cpu().numpy(), getattr(xx, name[3:])(axis=1)), (what, name)
        assert np.array_equal(i.cpu().numpy(), getattr(xx, arg)(axis=1)), (what, name, 'idx')
        assert np.array_equal(getattr(R, name)(xd, dim=1).cpu().numpy(), getattr(xx, name[3:])(axis=1)), (what, name)
    anyn = np.isnan(x).any(axis=1)
    v, i = R.min(xd, dim=1, return_indices=True)
    assert np.array_equal(np.isnan(v.cpu().numpy()), anyn), (what, 'min nan')
    ei = np.where(anyn, np.isnan(x).argmax(axis=1), np.nan_to_num(x, nan=np.inf).argmin(axis=1))
    assert np.array_equal(i.cpu().numpy(), ei), (what, 'min idx')
    big = max(1.0, float(np.nanmax(np.abs(x64)))) if np.isfinite(x64).any() else 1.0
    m, em = R.nanmean(xd, dim=1, dtype=torch.float64).cpu().numpy(), np.nanmean(x64, axis=1)
    assert np.array_equal(np.isnan(m), np.isnan(em)), (what, 'nanmean nan')
    assert np.nanmax(np.abs(m - em), initial=0) <= 1e-12 * big, (what, 'nanmean')
    for unb in (True, False):
        r = R.nanvar(xd, dim=1, unbiased=unb, dtype=torch.float64).cpu().numpy()
        ev = np.nanvar(x64, axis=1, ddof=int(unb))
        ok = np.isfinite(ev)
        assert np.array_equal(np.isnan(r[~ok]), np.isnan(ev[~ok])), (what, 'nanvar nan')
        assert np.abs(r[ok] - ev[ok]).max(initial=0) <= 1e-11 * big * big, (what, 'nanvar')
    s = R.std(xd, dim=1).cpu().numpy()
    assert np.array_equal(np.isnan(s), anyn | (x.shape[1] < 2)), (what, 'std nan')
    # median / nan-omitting median: the radix selection against a sort (torch on the CPU: lower median)
    xt = torch.from_numpy(x)
    for omit, fn in ((False, torch.median), (True, torch.nanmedian)):
        v, i = R.median(xd, dim=1, omitnan=omit, return_indices=True)
        v, i = v.cpu().numpy(), i.cpu().numpy()
        ev = fn(xt, dim=1).values.numpy()
        assert np.array_equal(np.isnan(v), np.isnan(ev)) and np.array_equal(v[~np.isnan(v)], ev[~np.isnan(ev)]), (what, 'median', omit)
        picked = np.take_along_axis(x, i[:, None, :], axis=1)[:, 0, :]
        assert np.array_equal(np.isnan(picked), np.isnan(v)) and np.array_equal(picked[~np.isnan(v)], v[~np.isnan(v)]), (what, 'median idx')


t0 = time.time()
n = 0
t_last = time.time()
while time.time() - t0 < budget:
    if time.time() - t_last > 45:      # keep the run visibly alive
        print(f'... {n} shapes', flush=True)
        t_last = time.time()
    o, r, i = pick_shape()
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    mean = float(rng.choice([0.0, 0.0, 1e3, -1e5]))
    x = (rng.standard_normal((o, r, i)) + mean).astype(dtype)
    pn = float(rng.choice([0.0, 0.02, 0.5]))
    if pn:
        x[rng.random(x.shape) < pn] = np.nan
    if rng.random() < 0.5:
        x[rng.random(x.shape) < 0.1] = dtype(mean + 0.5)   # ties
    off = int(rng.choice([0, 0, 1, 2, 3]))
    flat = np.concatenate([np.zeros(off, dtype), x.reshape(-1)])
    xd = torch.from_numpy(flat).to(dev)[off:].reshape(o, r, i)
    check(x, xd, (o, r, i, dtype.__name__, off, pn, mean))
    n += 1
print(f'fuzz_reduce: {n} shapes, 0 failures, {time.time() - t0:.0f} s')
