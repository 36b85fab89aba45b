#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r03h}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_qr.py tests/test_gpu_autograd.py -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_${TAG}.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
WHICH=qr timeout -k 10 300 python scripts/bench_reduce.py > $O/qr_table_${TAG}.md 2>/dev/null; echo "qr table rc=$?"; grep "vectors" $O/qr_table_${TAG}.md | grep f32
timeout -k 10 600 python scripts/bench_qr_large.py > $O/qr_large_table_${TAG}.md 2>/dev/null; echo "qr large rc=$?"; grep "vectors" $O/qr_large_table_${TAG}.md
