#!/bin/bash
# Which (dtype, order 9..16, operation) of the QR family compiles to a register kernel WITHOUT a private segment:
# the table behind `qr_large_fits` (nfm_qr.hip).  One hipcc run per combination with
# -Rpass-analysis=kernel-resource-usage; prints "N dtype op scratch= vgpr= spill= secs=" (spills with scratch=0
# are moves to the accumulation registers, not memory).  CPU only, ~25 minutes on 4 cores.
# usage: scripts/survey_qr_large.sh > profiles/r03/qr_large_register_fit.txt
cd "$(dirname "$0")"
one() {
  EN=$1; ET=$2; NAME=$3; OP=$4
  S=$(date +%s)
  OUT=$(/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-everything -I../nitorch_fastmath_amd/csrc -I../include \
        -DEN=$EN -DET=$ET "-DEOP=$OP" -mllvm -pragma-unroll-threshold=1000000 -Rpass-analysis=kernel-resource-usage \
        -x hip -c survey_qr_large_harness.hip -o /dev/null 2>&1)
  E=$(date +%s)
  SCR=$(echo "$OUT" | grep -o "ScratchSize \[bytes/lane\]: [0-9]*" | grep -o "[0-9]*$" | sort -n | tail -1)
  VG=$(echo "$OUT" | grep -o " VGPRs: [0-9]*" | grep -o "[0-9]*$" | sort -n | tail -1)
  SP=$(echo "$OUT" | grep -o "VGPRs Spill: [0-9]*" | grep -o "[0-9]*$" | sort -n | tail -1)
  echo "$EN $ET $NAME scratch=$SCR vgpr=$VG spill=$SP secs=$((E-S))"
}
export -f one
ops=("eig|EigSymOp<ET,EN,false,false>" "eigu|EigSymOp<ET,EN,true,false>" "eigf|EigSymOp<ET,EN,false,true>" "eiguf|EigSymOp<ET,EN,true,true>" "hess|HessOp<ET,EN,false,false>" "hessu|HessOp<ET,EN,false,true>" "hsym|HessOp<ET,EN,true,false>" "hsymu|HessOp<ET,EN,true,true>" "qr|QrHessOp<ET,EN>" "rq|RqHessOp<ET,EN,false>" "rqu|RqHessOp<ET,EN,true>")
for t in float double; do for n in 9 10 11 12 13 14 15 16; do for o in "${ops[@]}"; do
  echo "$n $t ${o%%|*} ${o#*|}"
done; done; done | xargs -P 4 -L 1 bash -c 'one "$0" "$1" "$2" "$3"'
