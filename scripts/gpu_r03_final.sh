#!/bin/bash
# round 3: full GPU suite, bench lines of every workload, rocprofv3 kernel traces of the same commands,
# SQ counters of the default (reference-order) eig kernels at 3x3 and 8x8, HBM traffic passes, tables.
# usage: gpu_r03_final.sh TAG [stage ...]   (stages: tests bench prof pmc tables fuzz; default all)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r03}
shift
STAGES=${*:-tests bench prof pmc tables fuzz}
has() { [[ " $STAGES " == *" $1 "* ]]; }
mkdir -p $O
cd $R
if has tests; then
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest_${TAG}.log | cut -c1-400
  [ $rc -eq 0 ] || exit $rc
  timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_${TAG}.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke_${TAG}.log
fi
if has bench; then
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_${TAG}_driver_shape.log 2>&1; echo "bench(driver flags) rc=$?"; tail -1 $O/bench_${TAG}_driver_shape.log | cut -c1-300
  for w in sym_solve4 sym_solve6 batchinv8 sym_invert3 eig3 eig8 nansum nanmax; do
    timeout -k 10 500 python bench.py --steps 100 --warmup 10 --workload $w > $O/bench_${TAG}_$w.log 2>&1; echo "bench $w rc=$?"; tail -1 $O/bench_${TAG}_$w.log | cut -c1-200
  done
  # channel-first (SoA) against AoS at the medium and the full batch (verdict item: the 17 % gap of the layouts table)
  for lay in aos soa; do for nn in 1.7e7 1e8; do for w in sym_solve4 sym_solve6; do
    timeout -k 10 300 python bench.py --steps 100 --warmup 10 --layout $lay --n $nn --no-cpu --workload $w > $O/bench_${TAG}_${w}_${lay}_${nn}.log 2>&1; echo "bench $w $lay $nn rc=$?"
  done; done; done
  timeout -k 10 300 python bench.py --gpus 3 --steps 20 --warmup 5 --n 2e7 --backend gloo --share-gpu > $O/bench_${TAG}_rehearsal3.log 2>&1; echo "rehearsal rc=$?"
fi
if has prof; then
  cd /tmp && export TMPDIR=/tmp
  for w in sym_solve4 sym_solve6 batchinv8 eig3 eig8 nansum; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$w -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu --workload $w > $O/rocprof_${TAG}_$w.log 2>&1; echo "rocprof $w rc=$?"
  done
  cd $R
  for w in sym_solve4 sym_solve6 batchinv8 eig3 eig8 nansum; do
    python scripts/summarize_prof.py $O/prof_${TAG}_$w $O/${w}_kernel_stats_${TAG}.md "${TAG} $w" > /dev/null; echo "summ $w rc=$?"
  done
fi
if has pmc; then
  cd /tmp && export TMPDIR=/tmp
  for w in eig3 eig8; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_${TAG}_${w}_a -- python3 $R/bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu --workload $w > $O/pmc_${TAG}_${w}_a.log 2>&1; echo "pmc $w a rc=$?"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${TAG}_${w}_b -- python3 $R/bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu --workload $w > $O/pmc_${TAG}_${w}_b.log 2>&1; echo "pmc $w b rc=$?"
  done
  # HBM traffic of the workloads whose roofline.traffic was null (separate FETCH_SIZE / WRITE_SIZE passes)
  for w in eig3 sym_invert3; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG}_${w}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu --workload $w > $O/pmc_${TAG}_${w}_fetch.log 2>&1; echo "pmc $w fetch rc=$?"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG}_${w}_write -- python3 $R/bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu --workload $w > $O/pmc_${TAG}_${w}_write.log 2>&1; echo "pmc $w write rc=$?"
  done
  cd $R
  python scripts/parse_sq.py $O/pmc_${TAG}_eig3_a $O/pmc_${TAG}_eig3_b "EigSymOp<float, 3, false, false>" $O/eig3_counters_${TAG}.json; echo "parse_sq eig3 rc=$?"
  python scripts/parse_sq.py $O/pmc_${TAG}_eig8_a $O/pmc_${TAG}_eig8_b "EigSymOp<float, 8, false, false>" $O/eig8_counters_${TAG}.json; echo "parse_sq eig8 rc=$?"
  python scripts/parse_pmc.py $O/pmc_${TAG}_eig3_fetch $O/pmc_${TAG}_eig3_write "EigSymOp<float, 3, false, false>" $O/traffic_eig3.json eig3 5e7 aos; echo "traffic eig3 rc=$?"
  python scripts/parse_pmc.py $O/pmc_${TAG}_sym_invert3_fetch $O/pmc_${TAG}_sym_invert3_write "InvertOp<double, 3, false>" $O/traffic_sym_invert3.json sym_invert3 1e5 aos; echo "traffic sym_invert3 rc=$?"
fi
if has tables; then
  WHICH=qr timeout -k 10 300 python scripts/bench_reduce.py > $O/qr_table_${TAG}.md 2>/dev/null; echo "qr table rc=$?"
  timeout -k 10 600 python scripts/bench_qr_large.py > $O/qr_large_table_${TAG}.md 2>/dev/null; echo "qr large table rc=$?"
  timeout -k 10 600 python scripts/bench_table.py > $O/throughput_table_${TAG}.md 2>/dev/null; echo "throughput table rc=$?"
  timeout -k 10 400 python scripts/bench_layouts.py > $O/layouts_table_${TAG}.md 2>/dev/null; echo "layouts rc=$?"
  timeout -k 10 400 python scripts/bench_median.py > $O/median_table_${TAG}.md 2>/dev/null; echo "median rc=$?"
  timeout -k 10 300 python scripts/bench_latency.py > $O/latency_table_${TAG}.md 2>/dev/null; echo "latency rc=$?"
  # no-exchange-first kernels of orders 9..16 against the pivoted kernels behind them, same box, same inputs
  timeout -k 10 300 python scripts/bench_spd_ab.py 2>/dev/null > $O/spd_ab_${TAG}_first.md; echo "spd ab (first) rc=$?"
  NFM_DEBUG=1 NFM_SPD_OFF=1 timeout -k 10 300 python scripts/bench_spd_ab.py 2>/dev/null > $O/spd_ab_${TAG}_pivoted.md; echo "spd ab (pivoted) rc=$?"
  NFM_DEBUG=1 NFM_SPD_OFF=1 ORDERS_SYM=9,12,16 ORDERS_GEN=9,12,16 timeout -k 10 400 python scripts/bench_table.py 2>/dev/null > $O/throughput_table_${TAG}_pivoted.md; echo "throughput (pivoted arm) rc=$?"
fi
if has fuzz; then
  timeout -k 10 400 python scripts/fuzz_gpu.py 150 51 > $O/fuzz_gpu_${TAG}.log 2>&1; echo "fuzz_gpu rc=$?"; tail -2 $O/fuzz_gpu_${TAG}.log
  timeout -k 10 400 python scripts/fuzz_reduce.py 120 52 > $O/fuzz_reduce_${TAG}.log 2>&1; echo "fuzz_reduce rc=$?"; tail -2 $O/fuzz_reduce_${TAG}.log
fi
