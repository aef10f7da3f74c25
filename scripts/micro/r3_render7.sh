#!/bin/bash
cd /root/repo
for lib in $LIBS; do
  for a in 1 6; do
    echo "== $lib IDE3D_CONV_ARITH=$a"
    export IDE3D_HIP_LIB=/root/repo/ide-3d_amd/$lib/libide3d_hip.so IDE3D_CONV_ARITH=$a
    timeout 300 python scripts/kernel_rooflines.py --only render_rays 2>&1 | grep -E "flops"
  done
done
IDE3D_HIP_LIB=/root/repo/ide-3d_amd/lib/libide3d_hip.so IDE3D_CONV_ARITH=6 timeout 600 python -m pytest tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -2
