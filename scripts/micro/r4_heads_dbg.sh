#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
for L in "$@"; do
  echo "== $L"
  IDE3D_HIP_LIB=$PWD/ide-3d_amd/$L/libide3d_hip.so timeout 120 python scripts/kernel_rooflines.py --iters 30 --only "dual head" 2>&1 | grep -E "dual head" | cut -c1-110
done
echo "== lib, old head kernel"
IDE3D_HEAD_NO_RESIDENT=1 timeout 120 python scripts/kernel_rooflines.py --iters 30 --only "dual head" 2>&1 | grep -E "dual head" | cut -c1-110
