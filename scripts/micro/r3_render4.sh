#!/bin/bash
cd /root/repo
for a in 1 6; do
  echo "== IDE3D_CONV_ARITH=$a"
  IDE3D_CONV_ARITH=$a timeout 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_shapes.py -m gpu -x -q 2>&1 | tail -3
  IDE3D_CONV_ARITH=$a timeout 300 python scripts/kernel_rooflines.py --only render_rays 2>&1 | grep -E "flops"
  IDE3D_CONV_ARITH=$a timeout 300 python scripts/kernel_rooflines.py --only sample_voxel 2>&1 | grep -E "flops|sample_voxel \["
done
