#!/bin/bash
# HBM traffic of the positive-definite-first kernels at 16x16 float32 (separate FETCH_SIZE / WRITE_SIZE passes)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_spd_fetch -- python3 $R/scripts/profile_spd.py > $O/pmc_spd_fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_spd_write -- python3 $R/scripts/profile_spd.py > $O/pmc_spd_write.log 2>&1; echo "write rc=$?"
cd $R
python scripts/parse_pmc.py $O/pmc_spd_fetch $O/pmc_spd_write "spd_kernel<float, 16, 0>" $O/traffic_spd_solve16.json "sym_solve 16x16 f32 (spd_kernel)" 1860119 aos
python scripts/parse_pmc.py $O/pmc_spd_fetch $O/pmc_spd_write "spd_kernel<float, 16, 1>" $O/traffic_spd_invert16.json "sym_invert 16x16 f32 (spd_kernel)" 1860119 aos
python scripts/parse_pmc.py $O/pmc_spd_fetch $O/pmc_spd_write "gen_kernel<float, 16, 4>" $O/traffic_gen_inv16.json "batchinv 16x16 f32 (gen_kernel)" 610351 aos
