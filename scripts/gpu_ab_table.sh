#!/bin/bash
# A/B of two builds of the library on one box: the throughput table with each
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-abt}
B=${2:-libnfm_hip_v1.so}
mkdir -p $O
cd $R
timeout -k 10 400 python scripts/bench_table.py > $O/tt_${TAG}_A.md 2>/dev/null; echo "A rc=$?"
NFM_DEBUG=1 NFM_HIP_LIB=$R/nitorch_fastmath_amd/$B timeout -k 10 400 python scripts/bench_table.py > $O/tt_${TAG}_B.md 2>/dev/null; echo "B rc=$?"
paste -d'|' <(grep " 9x9\|1[0-6]x1[0-6]" $O/tt_${TAG}_A.md | cut -d'|' -f2,7) <(grep " 9x9\|1[0-6]x1[0-6]" $O/tt_${TAG}_B.md | cut -d'|' -f7)
