#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() { local only="$1"; shift; echo "== [$only] $*"; env "$@" timeout 120 python scripts/kernel_rooflines.py --iters 20 --only "$only" 2>&1 | grep -E "bf16x6|f16x3" | cut -c1-120; }
run "transposed 3x3 512->256 in@64" A=1
run "transposed 3x3 512->256 in@64" IDE3D_MODCONV_SP_ROWS=8 IDE3D_MODCONV_SPLIT_MIN=256
run "transposed 3x3 512->256 in@64" IDE3D_MODCONV_SP_ROWS=8 IDE3D_MODCONV_SPLIT_MIN=256 IDE3D_MODCONV_NO_STRIP=1
run "transposed 3x3 256->128 in@128" A=1
run "transposed 3x3 256->128 in@128" IDE3D_MODCONV_SP_ROWS=4
