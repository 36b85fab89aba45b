#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02j}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "sym_solve 16 f64 2 0" "sym_solve 16 f64 2 1" "batchinv 16 f64 1 1" "batchinv 16 f64 2 0"; do
  set -- $cfg
  name=$1_$2_$3_r$4_l$5
  export NFM_ROWWAVE_ROWS=$4 NFM_ROWWAVE_LDS=$5
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_${TAG}_${name}_a -- python3 $R/scripts/rowwave_probe.py $1 $2 $3 > $O/pmc_${TAG}_${name}_a.log 2>&1; echo "pmc $name a rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${TAG}_${name}_b -- python3 $R/scripts/rowwave_probe.py $1 $2 $3 > $O/pmc_${TAG}_${name}_b.log 2>&1; echo "pmc $name b rc=$?"
  python3 $R/scripts/parse_sq.py $O/pmc_${TAG}_${name}_a $O/pmc_${TAG}_${name}_b "roww_kernel" $O/rowwave_counters_${TAG}_${name}.json
  python3 - <<PY
import json
d=json.load(open("$O/rowwave_counters_${TAG}_${name}.json"))
c=d['counters_per_launch']; w=c['SQ_WAVES']
print("$name", 'ns',d['kernel_ns_under_pmc'], {k: round(v/w,1) for k,v in c.items() if k!='SQ_WAVES'})
PY
done
