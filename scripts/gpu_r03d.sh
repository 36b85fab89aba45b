#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r03d}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_qr.py tests/test_gpu_reduce.py -m gpu -x -q -k "not full_size" > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest_${TAG}.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/bench_qr_large.py > $O/qr_large_table_${TAG}.md 2>$O/qr_large_${TAG}.err; echo "qr large rc=$?"; cat $O/qr_large_table_${TAG}.md
timeout -k 10 600 python scripts/bench_median.py > $O/median_table_${TAG}.md 2>$O/median_${TAG}.err; echo "median rc=$?"; cat $O/median_table_${TAG}.md
