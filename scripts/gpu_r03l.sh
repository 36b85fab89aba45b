#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r03l}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_reduce.py -m gpu -x -q -k median > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_${TAG}.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/bench_median.py > $O/median_table_${TAG}.md 2>/dev/null; echo "median rc=$?"; head -24 $O/median_table_${TAG}.md | tail -20
