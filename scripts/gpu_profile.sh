#!/bin/bash
# profile session: rocprofv3 kernel trace + separate PMC passes for the headline workload,
# bench lines for every config
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r01b}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in sym_solve4 sym_solve6 batchinv8 nansum; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$w -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu --workload $w > $O/rocprof_${TAG}_$w.log 2>&1; echo "rocprof $w rc=$?"
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG}_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
ls $O/pmc_${TAG}_fetch/* | head
cd $R
python scripts/parse_pmc.py $O/pmc_${TAG}_fetch $O/pmc_${TAG}_write "SolveOp<float, 4, 0>" $O/traffic_sym_solve4.json sym_solve4 1e8 aos
for w in sym_solve4 sym_solve6 batchinv8 sym_invert3 nansum nanmax; do
  timeout -k 10 400 python bench.py --steps 100 --warmup 10 --workload $w > $O/bench_${TAG}_$w.log 2>&1; echo "bench $w rc=$?"; tail -1 $O/bench_${TAG}_$w.log | cut -c1-330
done
# dim-wise reductions under the profiler (2^28 elements per shape keeps the trace short)
cd /tmp
WHICH=reduce LOG2N=28 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_reduce_dim -- python3 $R/scripts/bench_reduce.py > $O/rocprof_${TAG}_reduce_dim.log 2>&1; echo "rocprof reduce_dim rc=$?"
cd $R
WHICH=reduce timeout -k 10 300 python scripts/bench_reduce.py > $O/reduce_dim_table_${TAG}.md 2>/dev/null; echo "reduce table rc=$?"
timeout -k 10 300 python scripts/bench_layouts.py > $O/layouts_table_${TAG}.md 2>/dev/null; echo "layouts rc=$?"
timeout -k 10 300 python scripts/bench_latency.py > $O/latency_table_${TAG}.md 2>/dev/null; echo "latency rc=$?"
