#!/bin/bash
# round 2, session f: full GPU suite on the final library, bench lines of every workload, rocprofv3
# kernel traces of the same commands, SQ counters of the fast eig kernel, tables
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02w}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_${TAG}.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_${TAG}.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke_${TAG}.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_${TAG}_driver_shape.log 2>&1; echo "bench(driver flags) rc=$?"; tail -1 $O/bench_${TAG}_driver_shape.log | cut -c1-300
for w in sym_solve4 sym_solve6 batchinv8 sym_invert3 eig3 nansum nanmax; do
  timeout -k 10 500 python bench.py --steps 100 --warmup 10 --workload $w > $O/bench_${TAG}_$w.log 2>&1; echo "bench $w rc=$?"; tail -1 $O/bench_${TAG}_$w.log | cut -c1-200
done
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --layout soa --no-cpu > $O/bench_${TAG}_sym_solve4_soa.log 2>&1; echo "bench soa rc=$?"
cd /tmp && export TMPDIR=/tmp
for w in sym_solve4 sym_solve6 batchinv8 eig3 nansum; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$w -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu --workload $w > $O/rocprof_${TAG}_$w.log 2>&1; echo "rocprof $w rc=$?"
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_${TAG}_eig3_a -- python3 $R/bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu --workload eig3 > $O/pmc_${TAG}_eig3_a.log 2>&1; echo "pmc eig3 a rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${TAG}_eig3_b -- python3 $R/bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu --workload eig3 > $O/pmc_${TAG}_eig3_b.log 2>&1; echo "pmc eig3 b rc=$?"
cd $R
for w in sym_solve4 sym_solve6 batchinv8 eig3 nansum; do
  python scripts/summarize_prof.py $O/prof_${TAG}_$w $O/${w}_kernel_stats_${TAG}.md "${TAG} $w" > /dev/null; echo "summ $w rc=$?"
done
python scripts/parse_sq.py $O/pmc_${TAG}_eig3_a $O/pmc_${TAG}_eig3_b "EigSymOp<float, 3, false, true>" $O/eig3_counters_${TAG}.json; echo "parse_sq rc=$?"
WHICH=qr timeout -k 10 300 python scripts/bench_reduce.py > $O/qr_table_${TAG}.md 2>/dev/null; echo "qr table rc=$?"
timeout -k 10 600 python scripts/bench_table.py > $O/throughput_table_${TAG}.md 2>/dev/null; echo "throughput table rc=$?"
timeout -k 10 300 python scripts/bench_layouts.py > $O/layouts_table_${TAG}.md 2>/dev/null; echo "layouts rc=$?"
timeout -k 10 400 python scripts/fuzz_gpu.py 150 51 > $O/fuzz_gpu_${TAG}.log 2>&1; echo "fuzz_gpu rc=$?"; tail -2 $O/fuzz_gpu_${TAG}.log
timeout -k 10 400 python scripts/fuzz_reduce.py 120 52 > $O/fuzz_reduce_${TAG}.log 2>&1; echo "fuzz_reduce rc=$?"; tail -2 $O/fuzz_reduce_${TAG}.log
timeout -k 10 300 python scripts/bench_latency.py > $O/latency_table_${TAG}.md 2>/dev/null; echo "latency rc=$?"
timeout -k 10 300 python scripts/accuracy_study.py eig > $O/accuracy_eig_${TAG}.md 2>/dev/null; echo "acc rc=$?"
