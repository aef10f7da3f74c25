#!/bin/bash
# per-kernel A/B of library builds (directories under ide-3d_amd/) on the convolution rows of scripts/kernel_rooflines.py
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
ONLY=${ONLY:-conv}
for L in "$@"; do
  export IDE3D_HIP_LIB=$PWD/ide-3d_amd/$L/libide3d_hip.so
  echo "== $L"; python scripts/kernel_rooflines.py --iters 10 --only "$ONLY" 2>&1 | grep -v "^kernel case" 
done
