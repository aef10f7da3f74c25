#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() { local only="$1"; shift; echo "== [$only] $*"; env "$@" timeout 120 python scripts/kernel_rooflines.py --iters 20 --only "$only" 2>&1 | grep -E "bf16x6|f16x3|fp32" | cut -c1-120; }
run "32->128 in@128" A=1
run "32->128 in@128" IDE3D_MODCONV_SP_MINCIN=32
run "32->128 in@128" IDE3D_MODCONV_SP_MINCIN=32 IDE3D_MODCONV_SP_ROWS=4
