#!/usr/bin/env python
"""Orders 9..16: the lane-per-matrix register kernels (nfm_large.hip) against the
one-matrix-per-16-lanes kernels (nfm_rowwave.hip), both dtypes, every op they share.
The dispatch thresholds (nfm_rowwave.hip: rowwave_min_order) are read from the environment once
per process, so each arm runs in a child process.  Prints a markdown table.
usage: bench_rowwave.py            (parent: spawns the two arms)
       bench_rowwave.py --arm      (child: prints rows as 'name|n|bytes|ms')"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def arm():
    import torch
    import nitorch_fastmath_amd as N
    dev = torch.device('cuda:0')

    def timeit(fn, reps=6):
        fn()
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    for dtype, dn, sz in ((torch.float32, 'f32', 4), (torch.float64, 'f64', 8)):
        for M in range(9, 17):
            K = M * (M + 1) // 2
            n = int(1.2e9 / (2 * M * M * sz))
            g = torch.Generator(device=dev).manual_seed(M)
            mat = 0.3 * torch.randn(n, K, device=dev, generator=g, dtype=dtype) / M
            mat[:, :M] += 2
            vec = torch.randn(n, M, device=dev, generator=g, dtype=dtype)
            out = torch.empty_like(vec)
            inv = torch.empty_like(mat)
            print(f'sym_solve {M} {dn}|{n}|{(K + 2 * M) * sz}|{timeit(lambda: N.sym_solve(mat, vec, out=out)):.4f}')
            print(f'sym_invert {M} {dn}|{n}|{2 * K * sz}|{timeit(lambda: N.sym_invert(mat, out=inv)):.4f}')
            print(f'sym_det {M} {dn}|{n}|{(K + 1) * sz}|{timeit(lambda: N.sym_det(mat)):.4f}')
            del mat, vec, out, inv
            a = torch.randn(n, M, M, device=dev, generator=g, dtype=dtype) + 6 * torch.eye(M, device=dev, dtype=dtype)
            print(f'batchinv {M} {dn}|{n}|{2 * M * M * sz}|{timeit(lambda: N.batchinv(a)):.4f}')
            print(f'batchdet {M} {dn}|{n}|{(M * M + 1) * sz}|{timeit(lambda: N.batchdet(a)):.4f}')
            del a
            sys.stdout.flush()


def main():
    if '--arm' in sys.argv:
        return arm()
    res = {}
    for name, v in (('lane', '17'), ('row', '9')):
        env = dict(os.environ, NFM_ROWWAVE_MIN_F64=v, NFM_ROWWAVE_MIN_F32=v)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--arm'], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            sys.exit(1)
        for line in r.stdout.strip().split('\n'):
            if '|' in line:
                k, n, b, ms = line.split('|')
                res.setdefault(k, {})[name] = (int(n), int(b), float(ms))
    print('| op order dtype | batch | B/unit | lane-per-matrix ms | TB/s | 16-lanes-per-matrix ms | TB/s | row/lane speed-up |')
    print('|---|---|---|---|---|---|---|---|')
    for k, v in res.items():
        n, b, tl = v['lane']
        tr = v['row'][2]
        print(f'| {k} | {n:.2e} | {b} | {tl:.3f} | {n * b / tl / 1e9:.2f} | {tr:.3f} | {n * b / tr / 1e9:.2f} | {tl / tr:.2f} |')


if __name__ == '__main__':
    main()
