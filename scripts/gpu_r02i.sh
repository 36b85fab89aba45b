#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02i}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_qr.py tests/test_gpu_autograd.py tests/test_gpu_reference_cases.py tests/test_gpu_streams_graphs.py -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest_${TAG}.log | cut -c1-300
timeout -k 10 300 python scripts/accuracy_study.py eig > $O/accuracy_eig_${TAG}.md 2>/dev/null; echo "acc rc=$?"; cat $O/accuracy_eig_${TAG}.md
WHICH=qr timeout -k 10 300 python scripts/bench_reduce.py > $O/qr_table_${TAG}.md 2>/dev/null; echo "qr table rc=$?"; grep "eig_sym" $O/qr_table_${TAG}.md
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --workload eig3 > $O/bench_${TAG}_eig3.log 2>&1; echo "bench eig3 rc=$?"; tail -1 $O/bench_${TAG}_eig3.log | cut -c1-1200
