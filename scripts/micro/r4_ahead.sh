#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
IDE3D_HIP_LIB=$PWD/ide-3d_amd/lib_ah2/libide3d_hip.so timeout 400 python -m pytest tests/test_gpu_conv_arith.py -q -x -k "arithmetics_vs_float64 or strip_plan" 2>&1 | tail -2
for L in lib lib_ah2; do
  echo "== $L"
  IDE3D_HIP_LIB=$PWD/ide-3d_amd/$L/libide3d_hip.so timeout 200 python scripts/kernel_rooflines.py --iters 20 --only "modconv" 2>&1 | grep -E "bf16x6" | grep -v "32->128" | cut -c1-120
done
