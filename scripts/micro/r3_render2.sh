#!/bin/bash
# round 3: where the fused renderer's time goes — full kernel vs gathers only vs MLPs only, fp32 and split arithmetic
# (record of the experiment: the -DIDE3D_RM_NO_MLP / -DIDE3D_RM_NO_GATHER switches of raymarch.hip that built lib_nomlp / lib_nogather
#  existed only in that working tree; results in DESIGN.md 5.2c)
cd /root/repo
for lib in lib lib_nomlp lib_nogather; do
  for a in 1 6; do
    echo "== $lib IDE3D_CONV_ARITH=$a"
    IDE3D_HIP_LIB=/root/repo/ide-3d_amd/$lib/libide3d_hip.so IDE3D_CONV_ARITH=$a timeout 300 python scripts/kernel_rooflines.py --only render_rays 2>&1 | grep -E "MLP flops" 
  done
done
