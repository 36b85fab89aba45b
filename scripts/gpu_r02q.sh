#!/bin/bash
# round 2, session q: eig_sym fast sweeps (closed-form last stage): parity, bench eig3, qr table, accuracy
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02q}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_${TAG}.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --workload eig3 > $O/bench_${TAG}_eig3.log 2>&1; echo "bench eig3 rc=$?"; tail -1 $O/bench_${TAG}_eig3.log | cut -c1-400
WHICH=qr timeout -k 10 300 python scripts/bench_reduce.py > $O/qr_table_${TAG}.md 2>/dev/null; echo "qr table rc=$?"; grep "eig_sym" $O/qr_table_${TAG}.md
timeout -k 10 300 python scripts/accuracy_study.py eig > $O/accuracy_eig_${TAG}.md 2>/dev/null; echo "acc rc=$?"; cat $O/accuracy_eig_${TAG}.md | cut -c1-200
timeout -k 10 400 python scripts/fuzz_gpu.py 100 71 > $O/fuzz_gpu_${TAG}.log 2>&1; echo "fuzz_gpu rc=$?"; tail -2 $O/fuzz_gpu_${TAG}.log
