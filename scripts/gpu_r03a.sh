#!/bin/bash
# round 3, session a: qr tests on the new default arithmetic of eig_sym, qr table, eig3 bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r03a}
mkdir -p $O
cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_qr.py tests/test_abi_host.py tests/test_gpu_bench_contract.py -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest_${TAG}.log | cut -c1-400
[ $rc -eq 0 ] || exit $rc
WHICH=qr timeout -k 10 300 python scripts/bench_reduce.py > $O/qr_table_${TAG}.md 2>/dev/null; echo "qr table rc=$?"
grep eig_sym $O/qr_table_${TAG}.md
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --workload eig3 > $O/bench_${TAG}_eig3.log 2>&1; echo "bench eig3 rc=$?"; tail -1 $O/bench_${TAG}_eig3.log | cut -c1-1800
