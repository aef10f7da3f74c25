#!/bin/bash
# low-resolution 3x3 layers (4^2 .. 32^2): form / split-K sweep in the default arithmetic
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() { local only="$1"; shift; echo "== [$only] $*"; env "$@" timeout 120 python scripts/kernel_rooflines.py --iters 20 --only "$only" 2>&1 | grep -E "bf16x6" ; }
C32="modconv 3x3 512->512 @32"; C16="modconv 3x3 512->512 @16"; C8="modconv 3x3 512->512 @8"
T32="transposed 3x3 512->512 in@32"; T16="transposed 3x3 512->512 in@16"; T8="transposed 3x3 512->512 in@8"
run "$C32" A=1
run "$C32" IDE3D_MODCONV_SP_ROWS=16 IDE3D_MODCONV_SPLITK=4
run "$C32" IDE3D_MODCONV_SP_ROWS=16 IDE3D_MODCONV_SPLITK=8
run "$C32" IDE3D_SP_W8=3 IDE3D_MODCONV_SPLITK=4
run "$C32" IDE3D_SP_W8=3 IDE3D_MODCONV_SPLITK=8
run "$C32" IDE3D_MODCONV_SPLITK=4
run "$C32" IDE3D_MODCONV_SPLITK=8
run "$C16" A=1
run "$C16" IDE3D_SP_W8=3
run "$C16" IDE3D_SP_W8=3 IDE3D_MODCONV_SPLITK=16
run "$C16" IDE3D_MODCONV_SPLITK=16
run "$C16" IDE3D_MODCONV_SP_ROWS=16 IDE3D_MODCONV_SPLITK=16
run "$C8" A=1
run "$C8" IDE3D_MODCONV_SPLITK=16
run "$T16" A=1
run "$T16" IDE3D_MODCONV_SPLITK=2
run "$T16" IDE3D_MODCONV_SPLITK=4
run "$T16" IDE3D_MODCONV_SP_ROWS=8 IDE3D_MODCONV_SPLITK=4
run "$T16" IDE3D_MODCONV_SP_ROWS=8 IDE3D_MODCONV_SPLITK=8
run "$T8" A=1
run "$T8" IDE3D_MODCONV_SPLITK=16
