#!/usr/bin/env python
"""Randomised differential test: facade (HIP) vs CPU oracle over random ops, orders, dtypes,
batch shapes, broadcast patterns and memory layouts.  usage: fuzz_gpu.py [seconds] [seed]"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N  # noqa: E402
import oracle as O  # noqa: E402

dev = torch.device('cuda:0')
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TOL = {np.float32: 1e-6, np.float64: 1e-12}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(b).max() if b.size else 1.0
    return float(np.abs(a - b).max() / (d if d > 0 else 1.0)) if a.size else 0.0


def relayout(t, ncomp):
    """return a tensor with the same values and a random memory layout"""
    mode = rng.integers(0, 5)
    nd = t.dim()
    if mode == 0 or nd == ncomp:
        return t.contiguous()
    if mode == 1:      # component-major ("channel first")
        perm = list(range(nd - ncomp, nd)) + list(range(nd - ncomp))
        inv = np.argsort(perm).tolist()
        return t.permute(perm).contiguous().permute(inv)
    if mode == 2:      # strided batch (every other element of a bigger buffer)
        big = torch.zeros((t.shape[0] * 2,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        big[::2] = t
        return big[::2]
    if mode == 3:      # offset base pointer
        flat = torch.zeros(t.numel() + 3, dtype=t.dtype, device=t.device)
        flat[3:] = t.reshape(-1)
        return flat[3:].view(t.shape)
    perm = list(range(nd - ncomp))   # permuted batch dims
    rng.shuffle(perm)
    perm = perm + list(range(nd - ncomp, nd))
    inv = np.argsort(perm).tolist()
    return t.permute(perm).contiguous().permute(inv)


def rand_batch():
    k = rng.integers(0, 4)
    return tuple(int(x) for x in rng.integers(1, [1, 70, 9, 5][k] + 1, size=k)) if k else ()


def spd(batch, M, dtype):
    G = rng.standard_normal(batch + (M, M))
    A = G @ np.swapaxes(G, -1, -2) / M + np.eye(M)
    iu = [(i, j) for i in range(M) for j in range(i + 1, M)]
    cols = [A[..., i, i] for i in range(M)] + [A[..., i, j] for i, j in iu]
    return np.stack(cols, -1).astype(dtype)


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


count = fails = 0
t_end = time.time() + budget
t_last = time.time()
while time.time() < t_end:
    if time.time() - t_last > 45:      # keep the run visibly alive
        print(f'... {count} cases, {fails} failures', flush=True)
        t_last = time.time()
    dtype = [np.float32, np.float64][rng.integers(0, 2)]
    M = int(rng.integers(1, 17))
    op = ['solve', 'matvec', 'addmatvec', 'invert', 'invert_diag', 'det', 'batchinv', 'batchdet', 'batchmatvec', 'nansum',
          'nanmax', 'outer', 'to_full', 'eig', 'solve_inplace_eps', 'qr_misc', 'matmul'][rng.integers(0, 17)]
    batch = rand_batch()
    tol = TOL[dtype]
    try:
        if op in ('solve', 'matvec', 'addmatvec'):
            # independent broadcastable batch shapes for mat / vec
            bm = tuple(1 if rng.random() < 0.25 else s for s in batch)
            bv = tuple(1 if (rng.random() < 0.25 and m != 1) else s for s, m in zip(batch, bm))
            mat, vec = spd(bm, M, dtype), rng.standard_normal(bv + (M,)).astype(dtype)
            md, vd = relayout(T(mat), 1), relayout(T(vec), 1)
            if op == 'solve':
                got, ref = N.sym_solve(md, vd), O.sym_solve(mat, vec)
            elif op == 'matvec':
                got, ref, tol = N.sym_matvec(md, vd), O.sym_matvec(mat, vec), 0.0
            else:
                inp = rng.standard_normal(batch + (M,)).astype(dtype)
                got, ref, tol = N.sym_addmatvec(relayout(T(inp), 1), md, vd), O.sym_matvec(mat, vec, inp, 1), 0.0
        elif op == 'solve_inplace_eps':
            mat, vec = spd(batch, M, dtype), rng.standard_normal(batch + (M,)).astype(dtype)
            eps = [float(e) for e in rng.random(int(rng.integers(1, M + 1)))]
            e = np.array((eps + [eps[-1]] * M)[:M], dtype)
            mat_e = mat.copy()
            mat_e[..., :M] += e
            vd = relayout(T(vec), 1)
            r = N.sym_solve_(relayout(T(mat), 1), vd, eps=eps)
            assert r.data_ptr() == vd.data_ptr()
            got, ref = vd, O.sym_solve(mat_e, vec)
        elif op == 'matmul':
            k, d = int(rng.integers(1, 5)), int(rng.integers(1, 5))
            j = rng.standard_normal(batch + (k, d)).astype(dtype)
            h = spd(batch, k, dtype)
            got, ref, tol = N.sym_matmul(relayout(T(j), 2), relayout(T(h), 1)), O.sym_matmul(j, h), 0.0
        elif op == 'qr_misc':
            Mq = int(rng.integers(1, 17))
            a = rng.standard_normal(batch + (Mq, Mq)).astype(dtype)
            qtol = tol * max(1.0, Mq * Mq / 4.0) * 4
            which = rng.integers(0, 4)
            ad = relayout(T(a), 2)
            if which == 0:
                got, ref, tol = N.qr.hessenberg(ad), O.hessenberg(a), qtol
            elif which == 1:
                hz = np.triu(a, -1)
                q, r = N.qr.qr_hessenberg(relayout(T(hz), 2))
                qo, ro = O.qr_hessenberg(hz)
                got, ref, tol = torch.stack([q, r]), np.stack([qo, ro]), qtol
            elif which == 2:
                got, ref, tol = N.qr.hessenberg_sym(ad, upper=bool(rng.integers(0, 2))), None, qtol
                # recompute with the same `upper` is awkward to thread through: compare eigen-invariants
                s_in = np.triu(a) + np.swapaxes(np.triu(a, 1), -1, -2)
                g = got.cpu().numpy().astype(np.float64)
                ok1 = np.abs(np.trace(g, axis1=-2, axis2=-1) - np.trace(a, axis1=-2, axis2=-1)).max() <= 50 * qtol * max(1.0, np.abs(a).max()) * Mq
                count += 1
                if not ok1 or not np.allclose(g, np.swapaxes(g, -1, -2)):
                    fails += 1
                    print('FAIL hessenberg_sym trace/symmetry', dtype.__name__, Mq, batch)
                continue
            else:
                v = rng.standard_normal(batch + (Mq,)).astype(dtype)
                b = int(rng.integers(0, Mq))
                u, al = N.qr.householder(relayout(T(v), 1), basis=b, return_alpha=True)
                uo, alo = O.householder(v, b)
                got, ref, tol = torch.cat([u, al.unsqueeze(-1)], -1), np.concatenate([uo, alo[..., None]], -1), qtol
        elif op in ('invert', 'invert_diag', 'det', 'to_full'):
            mat = spd(batch, M, dtype)
            md = relayout(T(mat), 1)
            if op == 'invert':
                got, ref = N.sym_invert(md), O.sym_invert(mat)
            elif op == 'invert_diag':
                got, ref = N.sym_invert(md, diag=True), O.sym_invert(mat, diag=True)
            elif op == 'det':
                got, ref, tol = N.sym_det(md), O.sym_det(mat), tol * 4
            else:
                got, ref, tol = N.sym_to_full(md), O.sym_to_full(mat), 0.0
        elif op == 'outer':
            x = rng.standard_normal(batch + (M,)).astype(dtype)
            got, ref, tol = N.sym_outer(relayout(T(x), 1)), O.sym_outer(x), 0.0
        elif op in ('batchinv', 'batchdet'):
            a = (rng.standard_normal(batch + (M, M)) + 8 * np.eye(M)).astype(dtype)
            ad = relayout(T(a), 2)
            if op == 'batchinv':
                got, ref = N.batchinv(ad), O.batch_inv(a)
            else:
                got, ref, tol = N.batchdet(ad), O.batch_det(a), tol * 4
        elif op == 'batchmatvec':
            R, C = int(rng.integers(1, 9)), int(rng.integers(1, 9))
            a = rng.standard_normal(batch + (R, C)).astype(dtype)
            v = rng.standard_normal(batch + (C,)).astype(dtype)
            got, ref, tol = N.batchmatvec(relayout(T(a), 2), relayout(T(v), 1)), O.batch_matvec(a, v), 0.0
        elif op in ('nansum', 'nanmax'):
            shape = tuple(int(x) for x in rng.integers(1, 40, size=rng.integers(1, 4)))
            x = rng.standard_normal(shape).astype(dtype)
            x[rng.random(shape) < 0.1] = np.nan
            dims = None if rng.random() < 0.4 else sorted(set(int(d) for d in rng.integers(0, len(shape), size=rng.integers(1, len(shape) + 1))))
            xd = relayout(T(x), 0) if len(shape) > 1 else T(x)
            fn = N.reduce.nansum if op == 'nansum' else N.reduce.nanmax
            got = fn(xd, dim=dims)
            ref = O.reduce(op, x, dims)
            tol = 1e-5 if (op == 'nansum' and dtype == np.float32) else (1e-12 if op == 'nansum' else 0.0)
            if op == 'nansum':
                ref = ref.astype(np.float64)
                scale = np.nansum(np.abs(x))
                ok = np.abs(got.cpu().numpy().astype(np.float64) - ref).max() <= tol * max(scale, 1e-30) if ref.size else True
                count += 1
                if not ok:
                    fails += 1
                    print('FAIL', op, dtype.__name__, shape, dims)
                continue
        else:  # eig
            Mq = min(M, 8)
            a = rng.standard_normal(batch + (Mq, Mq)).astype(dtype)
            a = a + np.swapaxes(a, -1, -2)
            got = np.sort(N.eig_sym(relayout(T(a), 2)).cpu().numpy(), -1)
            ref = np.linalg.eigvalsh(a.astype(np.float64))
            tol = tol * max(1.0, Mq * Mq) * 8
            count += 1
            if rel(got, ref) > tol:
                fails += 1
                print('FAIL eig', dtype.__name__, Mq, batch, rel(got, ref))
            continue
        g = got.cpu().numpy()
        count += 1
        bad = g.shape != ref.shape or (not np.array_equal(g, ref) if tol == 0.0 else rel(g, ref) > tol)
        if bad:
            fails += 1
            print('FAIL', op, dtype.__name__, 'M', M, 'batch', batch, 'err', rel(g, ref) if g.shape == ref.shape else (g.shape, ref.shape))
    except Exception as e:  # noqa: BLE001
        fails += 1
        print('EXC', op, dtype.__name__, M, batch, repr(e)[:200])
print(f'fuzz: {count} cases, {fails} failures')
sys.exit(1 if fails else 0)
