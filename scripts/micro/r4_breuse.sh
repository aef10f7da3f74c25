#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_conv_arith.py -q -x -k "arithmetics_vs_float64 or strip_plan or epilogue_and_split_k or y_amax" 2>&1 | tail -3
for L in "$@"; do
  echo "== $L"
  IDE3D_HIP_LIB=$PWD/ide-3d_amd/$L/libide3d_hip.so timeout 200 python scripts/kernel_rooflines.py --iters 20 --only "transposed 3x3" 2>&1 | grep -E "bf16x6|f16x3" | grep -v "in@8\|in@4" | cut -c1-120
done
