#!/bin/bash
# round 3: where the fused renderer's time goes — full kernel vs gathers only vs MLPs only, fp32 and split arithmetic
cd /root/repo
for lib in lib lib_nomlp lib_nogather; do
  for a in 1 6; do
    echo "== $lib IDE3D_CONV_ARITH=$a"
    IDE3D_HIP_LIB=/root/repo/ide-3d_amd/$lib/libide3d_hip.so IDE3D_CONV_ARITH=$a timeout 300 python scripts/kernel_rooflines.py --only render_rays 2>&1 | grep -E "MLP flops" 
  done
done
