#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r03i}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_large_orders.py tests/test_gpu_sym.py -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_${TAG}.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/bench_layouts.py > $O/layouts_table_${TAG}.md 2>/dev/null; echo "layouts rc=$?"; grep "broadcast mat" $O/layouts_table_${TAG}.md
