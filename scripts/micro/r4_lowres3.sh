#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() { local only="$1"; shift; echo "== [$only] $*"; env "$@" timeout 120 python scripts/kernel_rooflines.py --iters 20 --only "$only" 2>&1 | grep -E "bf16x6|fp32" | cut -c1-120; }
run "modconv 3x3 512->512 @8" A=1
run "modconv 3x3 512->512 @8" IDE3D_MODCONV_TILE=0
run "modconv 3x3 512->512 @8" IDE3D_MODCONV_TILE=0 IDE3D_MODCONV_SPLITK=16
run "modconv 3x3 512->512 @8" IDE3D_MODCONV_TILE=0 IDE3D_SP_W8=3
run "modconv 3x3 512->512 @4" A=1
run "modconv 3x3 512->512 @4" IDE3D_MODCONV_TILE=0
run "modconv 3x3 512->512 @4" IDE3D_MODCONV_TILE=1
run "modconv 3x3 512->512 @4" IDE3D_MODCONV_TILE=0 IDE3D_MODCONV_SPLITK=16
