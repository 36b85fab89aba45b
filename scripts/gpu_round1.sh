#!/bin/bash
# one GPU-box session: parity tests, bench lines for every config, rocprof kernel trace
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for w in sym_solve4 sym_solve6 batchinv8 sym_invert3; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --workload $w > $O/bench_$w.log 2>&1; echo "bench $w rc=$?"; tail -1 $O/bench_$w.log | cut -c1-600
done
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --workload sym_solve4 --layout soa --no-cpu > $O/bench_sym_solve4_soa.log 2>&1; tail -1 $O/bench_sym_solve4_soa.log | cut -c1-400
for w in nansum nanmax; do
  timeout -k 10 400 python bench.py --steps 10 --warmup 2 --workload $w > $O/bench_$w.log 2>&1; echo "bench $w rc=$?"; tail -1 $O/bench_$w.log | cut -c1-600
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sym_solve4 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu > $O/rocprof_sym_solve4.log 2>&1; echo "rocprof rc=$?"
ls -R $O/prof_sym_solve4 | head -20
