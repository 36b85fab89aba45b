#!/bin/bash
# PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes) for the other bench workloads
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in sym_solve6 batchinv8 nansum nanmax; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc2_${w}_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --workload $w > $O/pmc2_${w}_$c.log 2>&1; echo "pmc $w $c rc=$?"
  done
done
cd $R
python scripts/parse_pmc.py $O/pmc2_sym_solve6_FETCH_SIZE $O/pmc2_sym_solve6_WRITE_SIZE "SolveOp<float, 6, 0>" $O/traffic_sym_solve6.json sym_solve6 1e8 aos | cut -c1-160
python scripts/parse_pmc.py $O/pmc2_batchinv8_FETCH_SIZE $O/pmc2_batchinv8_WRITE_SIZE "BatchInvOp<double, 8>" $O/traffic_batchinv8.json batchinv8 1e7 aos | cut -c1-160
python scripts/parse_pmc.py $O/pmc2_nansum_FETCH_SIZE $O/pmc2_nansum_WRITE_SIZE "reduce_all_k1<float, 0>" $O/traffic_nansum.json nansum 8589934592 aos | cut -c1-160
python scripts/parse_pmc.py $O/pmc2_nanmax_FETCH_SIZE $O/pmc2_nanmax_WRITE_SIZE "reduce_all_k1<float, 1>" $O/traffic_nanmax.json nanmax 8589934592 aos | cut -c1-160
