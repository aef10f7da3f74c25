#!/bin/bash
cd /root/repo
scripts/micro/bin/split_dot2_check
for a in 1 6; do
  echo "== IDE3D_CONV_ARITH=$a"
  export IDE3D_CONV_ARITH=$a
  timeout 900 python -m pytest tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -3
  timeout 300 python scripts/kernel_rooflines.py --only render_rays 2>&1 | grep -E "flops"
  timeout 300 python scripts/kernel_rooflines.py --only sample_voxel 2>&1 | grep -E "flops|sample_voxel \["
done
