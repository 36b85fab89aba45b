#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02k}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_reduce.py tests/test_gpu_streams_graphs.py tests/test_abi_host.py tests/test_gpu_autograd.py -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest_${TAG}.log | cut -c1-400
python - <<'PY'
import torch, time, sys
sys.path.insert(0, '.')
import nitorch_fastmath_amd as N
dev = torch.device('cuda:0')
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best
print('| shape (rows x red) | ours ms | GB/s of one pass | torch.median ms |')
print('|---|---|---|---|')
for rows, red in ((1, 1 << 30), (64, 1 << 24), (1 << 14, 1 << 16), (1 << 20, 1024), (1 << 22, 256), (1 << 24, 27), (1 << 25, 8)):
    x = torch.randn(rows, red, device=dev)
    t = timeit(lambda: N.reduce.median(x, dim=1))
    try:
        tt = timeit(lambda: torch.median(x, dim=1), reps=2)
    except Exception as e:
        tt = float('nan')
    print(f'| {rows} x {red} | {t*1e3:.3f} | {rows*red*4/t/1e9:.0f} | {tt*1e3:.3f} |')
    del x
PY
