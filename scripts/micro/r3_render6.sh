#!/bin/bash
cd /root/repo
for lib in lib lib_pft lib_vox3; do
  for a in 1 6; do
    echo "== $lib IDE3D_CONV_ARITH=$a"
    export IDE3D_HIP_LIB=/root/repo/ide-3d_amd/$lib/libide3d_hip.so IDE3D_CONV_ARITH=$a
    [ $lib != lib_vox3 ] && timeout 300 python scripts/kernel_rooflines.py --only render_rays 2>&1 | grep -E "flops"
    [ $lib != lib_pft ] && timeout 300 python scripts/kernel_rooflines.py --only sample_voxel 2>&1 | grep -E "flops|sample_voxel \["
  done
done
