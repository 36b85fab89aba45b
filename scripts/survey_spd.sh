#!/bin/bash
# register / scratch survey of the positive-definite-first kernels (nfm_spd.hip): every part in its own directory
R=/root/repo
PARTS=${*:-0 1 2 3 4 5 6 7}
for part in $PARTS; do
  d=$R/build/exp/spd/part$part; mkdir -p $d
  (cd $d && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-everything -I$R/nitorch_fastmath_amd/csrc -I$R/include -DNFM_SPD_PART=$part -c $R/nitorch_fastmath_amd/csrc/nfm_spd.hip -o spd.o -save-temps=obj 2>&1 | grep error -A5) &
done
wait
python3 - $PARTS <<'P'
import re, sys
for part in sys.argv[1:]:
    s=open(f'/root/repo/build/exp/spd/part{part}/nfm_spd-hip-amdgcn-amd-amdhsa-gfx950.s').read()
    for b in s.split('  - .agpr_count')[1:]:
        n=re.search(r'\.name:\s+(\S+)',b).group(1)
        m=re.search(r'(matvec_tiled|matvec_strided|spd_strided|spd|gen|redo)_kernelI(\w)Li(\d+)E(?:Li(\d)|Lb(\d))?',n)
        print(m.group(1)[-3:], m.group(2), m.group(3), 'op',m.group(4), 'vgpr',re.search(r'\.vgpr_count:\s+(\d+)',b).group(1), 'scratch',re.search(r'private_segment_fixed_size: (\d+)',b).group(1), 'spill',re.search(r'vgpr_spill_count: (\d+)',b).group(1))
P
