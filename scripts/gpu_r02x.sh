#!/bin/bash
# round 2, session x: MODE_PACKED (records contiguous inside, any batch stride): parity, fuzz, layouts table
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02x}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_${TAG}.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/bench_layouts.py > $O/layouts_table_${TAG}.md 2>$O/layouts_${TAG}.err; echo "layouts rc=$?"; cat $O/layouts_table_${TAG}.md
timeout -k 10 400 python scripts/fuzz_gpu.py 150 61 > $O/fuzz_gpu_${TAG}.log 2>&1; echo "fuzz_gpu rc=$?"; tail -2 $O/fuzz_gpu_${TAG}.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_${TAG}_driver_shape.log 2>&1; echo "bench(driver flags) rc=$?"; tail -1 $O/bench_${TAG}_driver_shape.log | cut -c1-300
