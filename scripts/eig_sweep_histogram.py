#!/usr/bin/env python
"""Per-matrix QR sweep counts of eig_sym (reference-order arithmetic = the CPU oracle's, whose
float32 iterates the 'reference' kernel reproduces bit for bit) and what they cost a 64-lane
wavefront that runs in lockstep: the divergence profile behind DESIGN.md's eig_sym section.
CPU only (oracle).  usage: eig_sweep_histogram.py > profiles/r02/eig_sweeps_histogram.md"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O   # noqa: E402

O.build()
rng = np.random.default_rng(0)
print('Input: symmetrised standard-normal matrices (the bench workload), 256 000 per row; sweeps are counted per '
      'active block size m (one sweep on an m x m block = m - 1 Givens rotations).\n')
print('| dtype | n | sweeps per matrix: mean | min | max | histogram of total sweeps (count per value from min) | '
      'rotations per matrix (useful) | rotations per matrix a lockstep wave pays | lane utilisation | '
      'rotations if converged lanes were refilled (every sweep at full order n) |')
print('|---|---|---|---|---|---|---|---|---|---|')
for dt in (np.float32, np.float64):
    for n in (3, 4, 6, 8):
        a = rng.standard_normal((64 * 4000, n, n)).astype(dt)
        a = a + a.transpose(0, 2, 1)
        _, s = O.eig_sym(a, return_sweeps=True)
        tot = s.sum(-1)
        w = s.reshape(-1, 64, n).astype(float)
        weight = np.arange(n)                      # block size m (index m - 1) costs m - 1 rotations
        useful = (s * weight).sum(-1).mean()
        lock = (w.max(1) * weight).sum(-1).mean()  # every stage runs until the slowest lane of the wave is done
        refill = (n - 1) * tot.mean()
        hist = np.bincount(tot)[tot.min():]
        print(f'| {dt.__name__} | {n} | {tot.mean():.2f} | {tot.min()} | {tot.max()} | {hist.tolist()} | {useful:.1f} | '
              f'{lock:.1f} | {useful / lock:.2f} | {refill:.1f} |')
print('\nReading: a wave pays for its slowest lane at every deflation stage (lane utilisation 0.62-0.71; the SQ counters of '
      'the kernel say 0.72-0.75: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)).  A persistent kernel that refills '
      'converged lanes cannot specialise its sweep on the block size any more -- lanes of one wave sit at different stages '
      '-- so every sweep costs n - 1 rotations; the last column is that cost with PERFECT refilling: equal to (n >= 6) or '
      'only 14 % below (n = 3) what lockstep already pays.  Lane refill was therefore not built; the speed-up came from '
      'the arithmetic of a sweep (DESIGN.md section 4).')
