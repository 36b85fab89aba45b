#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r04f}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_large_orders.py tests/test_gpu_sym.py tests/test_gpu_batched.py tests/test_gpu_autograd.py -m gpu -x -q > $O/pytest_${TAG}.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_${TAG}.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/bench_spd_ab.py 2>/dev/null > $O/spd_ab_${TAG}_first.md; echo "spd ab (first) rc=$?"
NFM_DEBUG=1 NFM_SPD_OFF=1 timeout -k 10 300 python scripts/bench_spd_ab.py 2>/dev/null > $O/spd_ab_${TAG}_pivoted.md; echo "spd ab (pivoted) rc=$?"
grep "indefinite" $O/spd_ab_${TAG}_first.md; grep "indefinite" $O/spd_ab_${TAG}_pivoted.md
