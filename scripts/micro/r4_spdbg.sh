#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
for L in "$@"; do
  echo "== $L (SP_DBG: 1 = weights fetched once, 2 = patch staged once, 4 = patch loaded once / committed every chunk, 8 = loaded every chunk / committed once)"
  IDE3D_HIP_LIB=$PWD/ide-3d_amd/$L/libide3d_hip.so timeout 200 python scripts/kernel_rooflines.py --iters 20 --only "modconv" 2>&1 | grep -E "bf16x6" | grep -v "in@8\|in@4\|@8 \|@16\|32->128\|@32" | cut -c1-120
done
