#!/bin/bash
# round 4: 8-wave forms of the split convolutions under exclusive residency (scripts/kernel_rooflines.py conv rows)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" python scripts/kernel_rooflines.py --iters 10 --only "modconv" 2>&1 | grep -E "f16x3|bf16x6" ; }
run IDE3D_SP_W8=0
run IDE3D_SP_W8=3
run IDE3D_SP_W8=3 IDE3D_MODCONV_SP_ROWS=16
run IDE3D_SP_W8=3 IDE3D_MODCONV_SP_ROWS=8
