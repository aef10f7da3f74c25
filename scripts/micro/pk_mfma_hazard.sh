#!/bin/bash
# Builds the reproducer of DESIGN.md section 4.2 twice (packed fp32 instructions on / off) and runs both on the GPU of this box.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
D=${1:-/tmp}
hipcc -O3 --offload-arch=gfx950 $R/scripts/micro/pk_mfma_hazard.cpp -o $D/hazard_pk
hipcc -O3 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops $R/scripts/micro/pk_mfma_hazard.cpp -o $D/hazard_nopk 2> >(grep -v "not a recognized feature" >&2)
echo "== victims compiled WITH packed fp32 instructions (v_pk_fma_f32)"; $D/hazard_pk || true
echo "== the same source compiled WITHOUT them (v_fma_f32 / v_fmac_f32)"; $D/hazard_nopk || true
