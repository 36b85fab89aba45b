#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r03c}
mkdir -p $O
cd $R
timeout -k 10 600 python scripts/bench_qr_large.py > $O/qr_large_table_${TAG}.md 2>$O/qr_large_${TAG}.err; echo "qr large rc=$?"; cat $O/qr_large_table_${TAG}.md
WHICH=qr timeout -k 10 300 python scripts/bench_reduce.py > $O/qr_table_${TAG}.md 2>/dev/null; echo "qr table rc=$?"
grep "eig_sym\|givens\|householder" $O/qr_table_${TAG}.md
timeout -k 10 300 python scripts/bench_layouts.py > $O/layouts_table_${TAG}.md 2>/dev/null; echo "layouts rc=$?"; cat $O/layouts_table_${TAG}.md
