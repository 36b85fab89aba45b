#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02h}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest_${TAG}.log | cut -c1-300
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_${TAG}.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke_${TAG}.log
timeout -k 10 300 python scripts/accuracy_study.py eig > $O/accuracy_eig_${TAG}.md 2>/dev/null; echo "acc rc=$?"; cat $O/accuracy_eig_${TAG}.md
WHICH=qr timeout -k 10 300 python scripts/bench_reduce.py > $O/qr_table_${TAG}.md 2>/dev/null; echo "qr table rc=$?"; grep "eig_sym\|givens" $O/qr_table_${TAG}.md
timeout -k 10 300 python scripts/bench_small_records.py > $O/small_records_${TAG}.md 2>/dev/null; echo "small rc=$?"; grep givens $O/small_records_${TAG}.md
