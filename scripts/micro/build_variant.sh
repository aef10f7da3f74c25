#!/bin/bash
# Build a variant of libide3d_hip.so that differs from ide-3d_amd/lib only in the given sources' compile flags:
#   scripts/micro/build_variant.sh <name> "<extra flags>" [file.hip ...]     (default file: triplane_tile.hip)
# -> ide-3d_amd/lib_<name>/libide3d_hip.so (git-ignored; travels to the GPU box).  The other objects are reused.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; EXTRA=$2; shift 2
FILES=${@:-triplane_tile.hip}
make -C $R/ide-3d_amd/csrc -j8 >/dev/null
D=$R/ide-3d_amd/lib_$NAME
mkdir -p $D/obj
cp -p $R/ide-3d_amd/lib/obj/*.o $D/obj/
for f in $FILES; do rm -f $D/obj/${f%.hip}.o; done
make -C $R/ide-3d_amd/csrc -j8 OUTDIR=../lib_$NAME OBJDIR=../lib_$NAME/obj EXTRA="$EXTRA" 2>&1 | grep -E "error|Spill" || true
ls -la $D/libide3d_hip.so
