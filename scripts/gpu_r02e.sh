#!/bin/bash
# round 2, session e: row-wave kernels, pivot-row broadcast through LDS vs ds_bpermute
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02e}
mkdir -p $O
cd $R
for rows in 1 2 4; do
  NFM_ROWWAVE_LDS=1 NFM_ROWWAVE_MIN_F32=9 NFM_ROWWAVE_MIN_F64=9 NFM_ROWWAVE_ROWS=$rows timeout -k 10 600 python -m pytest tests/test_gpu_large_orders.py -x -q > $O/pytest_${TAG}_rows$rows.log 2>&1; rc=$?; echo "pytest lds rows=$rows rc=$rc"; tail -3 $O/pytest_${TAG}_rows$rows.log | cut -c1-300
  [ $rc -eq 0 ] || exit $rc
done
timeout -k 10 900 python scripts/bench_rowwave.py > $O/rowwave_table_${TAG}.md 2> $O/rowwave_${TAG}.err; echo "rowwave rc=$?"; cat $O/rowwave_table_${TAG}.md; tail -5 $O/rowwave_${TAG}.err
