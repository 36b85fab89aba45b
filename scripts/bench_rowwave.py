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

    from _timing import timeit as _timeit

    def timeit(fn, reps=6):
        return _timeit(fn, reps) * 1e3     # ms, steady state (scripts/_timing.py)


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
    # (name, minimum order of the row-wave path, rows per lane, pivot row broadcast through LDS)
    arms = (('lane', '17', '0', '0'), ('r1 bperm', '9', '1', '0'), ('r2 bperm', '9', '2', '0'), ('r4 bperm', '9', '4', '0'),
            ('r1 lds', '9', '1', '1'), ('r2 lds', '9', '2', '1'), ('r4 lds', '9', '4', '1'))
    for name, v, rows, lds in arms:
        env = dict(os.environ, NFM_DEBUG='1', NFM_ROWWAVE_MIN_F64=v, NFM_ROWWAVE_MIN_F32=v, NFM_ROWWAVE_ROWS=rows, NFM_ROWWAVE_LDS=lds)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--arm'], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            sys.exit(1)
        for line in r.stdout.strip().split('\n'):
            if '|' in line:
                k, n, b, ms = line.split('|')
                res.setdefault(k, {})[name] = (int(n), int(b), float(ms))
    names = [a[0] for a in arms]
    print('TB/s of algorithmic bytes; lane = one matrix per lane (nfm_large.hip); rK = K rows per lane, 16/K lanes per '
          'matrix (nfm_rowwave.hip), pivot row broadcast by ds_bpermute or through an LDS slot\n')
    print('| op order dtype | batch | B/unit | ' + ' | '.join(names) + ' | best |')
    print('|---|---|---|' + '---|' * (len(names) + 1))
    for k, v in res.items():
        n, b, _ = v['lane']
        tb = {a: n * b / v[a][2] / 1e9 for a in v}
        best = max(tb, key=tb.get)
        print(f"| {k} | {n:.2e} | {b} | " + ' | '.join(f'{tb[a]:.2f}' for a in names) + f' | {best} |')


if __name__ == '__main__':
    main()
