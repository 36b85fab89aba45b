#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02g}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_large_orders.py tests/test_gpu_sym.py tests/test_gpu_multi_device.py tests/test_gpu_bench_contract.py -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest_${TAG}.log | cut -c1-300
timeout -k 10 300 python scripts/bench_small_records.py > $O/small_records_${TAG}.md 2>$O/small_records_${TAG}.err; echo "small rc=$?"; cat $O/small_records_${TAG}.md; tail -3 $O/small_records_${TAG}.err
for w in batchinv8 nansum sym_invert3; do
  timeout -k 10 500 python bench.py --steps 30 --warmup 5 --workload $w > $O/bench_${TAG}_$w.log 2>&1; echo "bench $w rc=$?"; tail -1 $O/bench_${TAG}_$w.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cpu_baseline'])"
done
