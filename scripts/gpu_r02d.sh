#!/bin/bash
# round 2, session d: rows-per-lane sweep of the row-wave kernels, unaligned-base TileIO A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02d}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_large_orders.py tests/test_gpu_sym.py tests/test_gpu_batched.py -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest_${TAG}.log | cut -c1-400
for rows in 1 4; do
  NFM_ROWWAVE_ROWS=$rows timeout -k 10 600 python -m pytest tests/test_gpu_large_orders.py -x -q > $O/pytest_${TAG}_rows$rows.log 2>&1; echo "pytest rows=$rows rc=$?"; tail -3 $O/pytest_${TAG}_rows$rows.log | cut -c1-300
done
timeout -k 10 900 python scripts/bench_rowwave.py > $O/rowwave_table_${TAG}.md 2> $O/rowwave_${TAG}.err; echo "rowwave rc=$?"; cat $O/rowwave_table_${TAG}.md; tail -5 $O/rowwave_${TAG}.err
timeout -k 10 300 python scripts/bench_layouts.py > $O/layouts_table_${TAG}.md 2>/dev/null; echo "layouts rc=$?"; cat $O/layouts_table_${TAG}.md
NFM_TILE_ALIGN16=1 timeout -k 10 300 python scripts/bench_layouts.py > $O/layouts_table_${TAG}_align16.md 2>/dev/null; echo "layouts(align16) rc=$?"; grep "off by\|x\[1:\]\|contiguous AoS" $O/layouts_table_${TAG}_align16.md
