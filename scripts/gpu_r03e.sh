#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r03e}
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/test_gpu_qr.py tests/test_gpu_reduce.py tests/test_gpu_large_orders.py tests/test_gpu_sym.py tests/test_gpu_batched.py tests/test_gpu_streams_graphs.py -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest_${TAG}.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/bench_qr_large.py > $O/qr_large_table_${TAG}.md 2>$O/qr_large_${TAG}.err; echo "qr large rc=$?"; grep -v " 8x8 " $O/qr_large_table_${TAG}.md
timeout -k 10 600 python scripts/bench_median.py > $O/median_table_${TAG}.md 2>$O/median_${TAG}.err; echo "median rc=$?"; head -26 $O/median_table_${TAG}.md
timeout -k 10 400 python scripts/bench_layouts.py > $O/layouts_table_${TAG}.md 2>/dev/null; echo "layouts rc=$?"; grep "| 12 |\|| 16 |" $O/layouts_table_${TAG}.md
ORDERS_SYM=12,16 ORDERS_GEN=12,16 timeout -k 10 600 python scripts/bench_table.py > $O/throughput_table_${TAG}.md 2>/dev/null; echo "throughput table rc=$?"; grep -v eig_sym $O/throughput_table_${TAG}.md
timeout -k 10 900 python scripts/bench_rowwave.py > $O/rowwave_table_${TAG}.md 2>$O/rowwave_${TAG}.err; echo "rowwave rc=$?"; tail -5 $O/rowwave_table_${TAG}.md
