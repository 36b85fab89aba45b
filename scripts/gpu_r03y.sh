#!/bin/bash
# no-exchange-first kernels (nfm_spd.hip): parity of the large orders, then the throughput rows of both families
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r03y}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_large_orders.py tests/test_gpu_batched.py tests/test_gpu_sym.py -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_${TAG}.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
ORDERS_SYM=${ORDERS_SYM:-9,12,14,16} ORDERS_GEN=${ORDERS_GEN:-9,12,14,16} timeout -k 10 500 python scripts/bench_table.py > $O/tt_${TAG}.md 2>&1; echo "table rc=$?"; grep "batch\|sym_solve\|sym_invert" $O/tt_${TAG}.md
