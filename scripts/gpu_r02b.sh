#!/bin/bash
# round 2, session b: fast float32 sweeps of eig_sym -- accuracy tables, QR tests, bench + counters
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02b}
mkdir -p $O
cd $R
for w in eig large qr; do
  timeout -k 10 300 python scripts/accuracy_study.py $w > $O/accuracy_${w}_${TAG}.md 2> $O/accuracy_${w}_${TAG}.err; echo "accuracy $w rc=$?"; cat $O/accuracy_${w}_${TAG}.md
done
timeout -k 10 600 python -m pytest tests/test_gpu_qr.py tests/test_gpu_autograd.py tests/test_gpu_reference_cases.py -q > $O/pytest_qr_${TAG}.log 2>&1; echo "pytest qr rc=$?"; tail -15 $O/pytest_qr_${TAG}.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --workload eig3 > $O/bench_${TAG}_eig3.log 2>&1; echo "bench eig3 rc=$?"; tail -1 $O/bench_${TAG}_eig3.log | cut -c1-1500
WHICH=qr timeout -k 10 300 python scripts/bench_reduce.py > $O/qr_table_${TAG}.md 2>/dev/null; echo "qr table rc=$?"; cat $O/qr_table_${TAG}.md
