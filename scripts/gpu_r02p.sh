#!/bin/bash
# final profile refresh: eig_sym (bench line, rocprofv3 trace, SQ counters), qr / throughput tables, accuracy
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02p}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_${TAG}.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --workload eig3 > $O/bench_${TAG}_eig3.log 2>&1; echo "bench eig3 rc=$?"; tail -1 $O/bench_${TAG}_eig3.log | cut -c1-200
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_${TAG}_driver_shape.log 2>&1; echo "bench(driver flags) rc=$?"; tail -1 $O/bench_${TAG}_driver_shape.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_eig3 -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu --workload eig3 > $O/rocprof_${TAG}_eig3.log 2>&1; echo "rocprof eig3 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_${TAG}_eig3_a -- python3 $R/bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu --workload eig3 > $O/pmc_${TAG}_eig3_a.log 2>&1; echo "pmc eig3 a rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${TAG}_eig3_b -- python3 $R/bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu --workload eig3 > $O/pmc_${TAG}_eig3_b.log 2>&1; echo "pmc eig3 b rc=$?"
cd $R
python scripts/summarize_prof.py $O/prof_${TAG}_eig3 $O/eig3_kernel_stats_${TAG}.md "${TAG} eig3" > /dev/null
python scripts/parse_sq.py $O/pmc_${TAG}_eig3_a $O/pmc_${TAG}_eig3_b "EigSymOp<float, 3, false, true>" $O/eig3_counters_${TAG}.json; echo "parse_sq rc=$?"
timeout -k 10 300 python scripts/accuracy_study.py eig > $O/accuracy_eig_${TAG}.md 2>/dev/null; echo "acc rc=$?"
WHICH=qr timeout -k 10 300 python scripts/bench_reduce.py > $O/qr_table_${TAG}.md 2>/dev/null; echo "qr table rc=$?"
timeout -k 10 600 python scripts/bench_table.py > $O/throughput_table_${TAG}.md 2>/dev/null; echo "throughput table rc=$?"
