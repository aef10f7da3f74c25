#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" python scripts/kernel_rooflines.py --iters 10 --only "low-res" 2>&1 | grep -E "f16x3|bf16x6|fp32" ; }
run IDE3D_HIP_LIB=$PWD/ide-3d_amd/lib_shared/libide3d_hip.so
run IDE3D_SP_W8=2
run IDE3D_SP_W8=3 IDE3D_MODCONV_SP_ROWS=8
