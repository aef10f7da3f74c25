#!/bin/bash
# round 3: decoder MLPs on the bf16 matrix path — render tests under each arithmetic, then the bench
cd /root/repo
mkdir -p gpurun_out
for a in 1 6 16; do
  echo "== IDE3D_CONV_ARITH=$a"
  IDE3D_CONV_ARITH=$a timeout 600 python -m pytest tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -3
done
timeout 600 python -m pytest tests/test_gpu_bench_config.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 30 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench_render1.json
timeout 300 python scripts/kernel_rooflines.py 2>&1 | grep -iE "render|voxel|lattice|cube" | head
