#!/bin/bash
# strided positive-definite-first kernels: parity (large orders, sym, autograd), then the layouts table
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r04c}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_large_orders.py tests/test_gpu_sym.py tests/test_gpu_batched.py tests/test_gpu_autograd.py tests/test_gpu_streams_graphs.py -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_${TAG}.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python scripts/bench_layouts.py > $O/layouts_${TAG}.md 2>/dev/null; echo "layouts rc=$?"; grep "| 12 |\|| 16 |" $O/layouts_${TAG}.md
