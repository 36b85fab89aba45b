#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r04i}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_large_orders.py tests/test_gpu_sym.py tests/test_gpu_autograd.py tests/test_gpu_streams_graphs.py -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_${TAG}.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
python - <<'P'
import sys, os, torch
sys.path.insert(0, 'scripts')
import nitorch_fastmath_amd as N
from _timing import timeit
dev = torch.device('cuda:0')
print('| sym_matvec, channel-first storage | M | batch | ms | GB/s |')
for M in (12, 16):
    K = M * (M + 1) // 2
    n = 4_250_000
    g = torch.Generator(device=dev).manual_seed(M)
    mat = torch.randn(K, n, device=dev, generator=g).t()
    vec = torch.randn(M, n, device=dev, generator=g).t()
    t = timeit(lambda: N.sym_matvec(mat, vec), 6)
    print(f'| float32 | {M} | {n:.2e} | {t*1e3:.3f} | {n*(K+2*M)*4/t/1e9:.0f} |')
    matc, vecc = mat.contiguous(), vec.contiguous()
    t = timeit(lambda: N.sym_matvec(matc, vecc), 6)
    print(f'| float32 contiguous | {M} | {n:.2e} | {t*1e3:.3f} | {n*(K+2*M)*4/t/1e9:.0f} |')
P
