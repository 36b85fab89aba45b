#!/usr/bin/env python
"""Error of the GPU kernels and of the CPU oracle against an fp64 truth (numpy.linalg in float64
on the SAME inputs), per op / order / dtype -- the numbers behind the error models asserted in
tests/test_gpu_qr.py and tests/test_gpu_large_orders.py.  Prints a markdown table.
usage: accuracy_study.py [eig|qr|large]   (GPU box)"""
import sys
import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import nitorch_fastmath_amd as N   # noqa: E402
import oracle as O                 # noqa: E402

dev = torch.device('cuda:0')
EPS = {np.float32: 2.0 ** -23, np.float64: 2.0 ** -52}


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def eig():
    print('| dtype | n | gpu vs oracle (sorted) | gpu vs eigvalsh64 | oracle vs eigvalsh64 | ratio gpu/oracle | same order frac | in n*eps |')
    print('|---|---|---|---|---|---|---|---|')
    for dtype in (np.float32, np.float64):
        for n in (2, 3, 4, 5, 6, 8, 12, 16):
            nb = 20000 if n <= 8 else 2000
            rng = np.random.default_rng(1000 + n)
            a = rng.standard_normal((nb, n, n)).astype(dtype)
            sym = ((a + a.transpose(0, 2, 1)) / 2).astype(dtype)
            ev = N.qr.eig_sym(t(sym)).cpu().numpy()
            ref = O.eig_sym(sym)
            truth = np.linalg.eigvalsh(sym.astype(np.float64))
            eg, eo = rel(np.sort(ev, -1), truth), rel(np.sort(ref, -1), truth)
            same = float(np.mean(np.abs(ev - ref).max(-1) <= 64 * EPS[dtype] * np.abs(ref).max()))
            print(f'| {dtype.__name__} | {n} | {rel(np.sort(ev, -1), np.sort(ref, -1)):.2e} | {eg:.2e} | {eo:.2e} | '
                  f'{eg / eo:.2f} | {same:.4f} | {eg / (n * EPS[dtype]):.2f} |')


def large():
    print('| dtype | n | subset | cond(A) max | inv: gpu vs oracle | inv: gpu vs inv64 | oracle vs inv64 | eps*cond | det: gpu vs oracle | det gpu vs det64 | oracle vs det64 |')
    print('|---|---|---|---|---|---|---|---|---|---|---|')
    for dtype in (np.float32, np.float64):
        for n in (9, 12, 14, 16):
            nb = 2000 + n
            rng = np.random.default_rng(70 + n)
            a = (rng.standard_normal((nb, n, n)) + 8 * np.eye(n)).astype(dtype)
            a[::7, 0, 0] = 0
            easy = np.ones(nb, bool)
            easy[::7] = False
            gi = N.batched.batchinv(t(a)).cpu().numpy()
            gd = N.batched.batchdet(t(a)).cpu().numpy()
            oi, od = O.batch_inv(a), O.batch_det(a)
            a64 = a.astype(np.float64)
            ti, td = np.linalg.inv(a64), np.linalg.det(a64)
            cond = np.linalg.cond(a64)
            for name, m in (('easy', easy), ('zero-pivot', ~easy)):
                # per-matrix relative error (max-norm of the matrix), worst over the subset
                def pm(x, y):
                    x, y = x[m].astype(np.float64), y[m].astype(np.float64)
                    return float((np.abs(x - y).reshape(len(x), -1).max(-1) / np.abs(y).reshape(len(y), -1).max(-1)).max())
                print(f'| {dtype.__name__} | {n} | {name} | {cond[m].max():.1f} | {pm(gi, oi):.2e} | {pm(gi, ti):.2e} | '
                      f'{pm(oi, ti):.2e} | {EPS[dtype] * cond[m].max():.2e} | {pm(gd[:, None], od[:, None]):.2e} | '
                      f'{pm(gd[:, None], td[:, None]):.2e} | {pm(od[:, None], td[:, None]):.2e} |')


def qr():
    print('| dtype | n | op | gpu vs oracle | gpu vs f64-oracle | oracle vs f64-oracle |')
    print('|---|---|---|---|---|---|')
    for dtype in (np.float32,):
        for n in (3, 5, 8, 12, 16):
            nb = 777 if n <= 8 else 130
            rng = np.random.default_rng(1000 + n)
            a = rng.standard_normal((nb, n, n)).astype(dtype)
            a64 = a.astype(np.float64)
            ops = {
                'hessenberg': (lambda x: N.qr.hessenberg(t(x)).cpu().numpy(), lambda x: O.hessenberg(x, False)),
                'hessenberg_sym': (lambda x: N.qr.hessenberg_sym(t(x), upper=True).cpu().numpy(),
                                   lambda x: O.hessenberg_sym(x, True, True)),
                'qr_hessenberg.q': (lambda x: N.qr.qr_hessenberg(t(np.triu(x, -1)))[0].cpu().numpy(),
                                    lambda x: O.qr_hessenberg(np.triu(x, -1))[0]),
                'qr_hessenberg.r': (lambda x: N.qr.qr_hessenberg(t(np.triu(x, -1)))[1].cpu().numpy(),
                                    lambda x: O.qr_hessenberg(np.triu(x, -1))[1]),
                'rq_hessenberg': (lambda x: N.qr.rq_hessenberg(t(np.triu(x, -1))).cpu().numpy(),
                                  lambda x: O.rq_hessenberg(np.triu(x, -1))),
            }
            for name, (g, o) in ops.items():
                got, ora, tru = g(a), o(a), o(a64)
                if isinstance(ora, tuple):
                    ora, tru = ora[0], tru[0]
                print(f'| {dtype.__name__} | {n} | {name} | {rel(got, ora):.2e} | {rel(got, tru):.2e} | {rel(ora, tru):.2e} |')


if __name__ == '__main__':
    O.build()
    which = sys.argv[1] if len(sys.argv) > 1 else 'eig'
    {'eig': eig, 'large': large, 'qr': qr}[which]()
