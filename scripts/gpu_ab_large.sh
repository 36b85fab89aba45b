#!/bin/bash
# A/B of two builds on one box: every op / order 9..16 / dtype of the large-order kernels (shipping dispatch)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-abl}
B=${2:-libnfm_hip_v1.so}
mkdir -p $O
cd $R
for rep in 1 2; do
  timeout -k 10 300 python scripts/bench_rowwave.py --arm > $O/lg_${TAG}_A$rep.txt 2>/dev/null; echo "A$rep rc=$?"
  NFM_DEBUG=1 NFM_HIP_LIB=$R/nitorch_fastmath_amd/$B timeout -k 10 300 python scripts/bench_rowwave.py --arm > $O/lg_${TAG}_B$rep.txt 2>/dev/null; echo "B$rep rc=$?"
done
python - <<P
import sys
def rd(f): return {l.split('|')[0]: float(l.strip().split('|')[3]) for l in open(f) if '|' in l}
O='$O'; T='$TAG'
a1,a2,b1,b2=[rd(f'{O}/lg_{T}_{x}.txt') for x in ('A1','A2','B1','B2')]
for k in a1:
    a=min(a1[k],a2[k]); b=min(b1[k],b2[k])
    flag = 'B' if b < 0.95*a else ('A' if a < 0.95*b else '')
    print(f'{k:22s} A {a:.4f} B {b:.4f} ratio {a/b:.2f} {flag}')
P
