#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" python scripts/kernel_rooflines.py --iters 20 --only "transposed 3x3 512->512 in@32" 2>&1 | grep -E "bf16x6|f16x3" ; }
run A=1
run IDE3D_MODCONV_TA_OLD=1
run IDE3D_MODCONV_TA_OLD=1 IDE3D_MODCONV_SP_ROWS=8
run IDE3D_MODCONV_SP_ROWS=8
run IDE3D_MODCONV_SP_ROWS=16
