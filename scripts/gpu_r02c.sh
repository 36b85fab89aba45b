#!/bin/bash
# round 2, session c: row-wave kernels (orders 9..16, one matrix per 16 lanes) + both eig arithmetic modes
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02c}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_large_orders.py tests/test_gpu_qr.py tests/test_gpu_autograd.py tests/test_gpu_reduce.py tests/test_gpu_reference_cases.py -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest_${TAG}.log | cut -c1-400
timeout -k 10 600 python scripts/bench_rowwave.py > $O/rowwave_table_${TAG}.md 2> $O/rowwave_${TAG}.err; echo "rowwave rc=$?"; cat $O/rowwave_table_${TAG}.md; tail -5 $O/rowwave_${TAG}.err
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --workload eig3 --no-cpu > $O/bench_${TAG}_eig3.log 2>&1; echo "bench eig3 rc=$?"; tail -1 $O/bench_${TAG}_eig3.log | cut -c1-300
