#!/bin/bash
# A/B of two builds of the library on one box: the qr table with each, interleaved twice
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-ab}
B=${2:-libnfm_hip_v1.so}
mkdir -p $O
cd $R
for rep in 1 2; do
  WHICH=qr timeout -k 10 300 python scripts/bench_reduce.py > $O/qr_${TAG}_A$rep.md 2>/dev/null; echo "A$rep rc=$?"
  NFM_DEBUG=1 NFM_HIP_LIB=$R/nitorch_fastmath_amd/$B WHICH=qr timeout -k 10 300 python scripts/bench_reduce.py > $O/qr_${TAG}_B$rep.md 2>/dev/null; echo "B$rep rc=$?"
done
paste -d'|' <(grep eig_sym $O/qr_${TAG}_A1.md | grep -v reference | cut -d'|' -f2,5) <(grep eig_sym $O/qr_${TAG}_A2.md | grep -v reference | cut -d'|' -f5) <(grep eig_sym $O/qr_${TAG}_B1.md | grep -v reference | cut -d'|' -f5) <(grep eig_sym $O/qr_${TAG}_B2.md | grep -v reference | cut -d'|' -f5)
