#!/bin/bash
# A/B of several builds of the library inside ONE gpurun call (box-to-box variance is larger than most kernel changes):
#   scripts/ab_libs.sh lib_a lib_b ...     (directories under ide-3d_amd/, each holding libide3d_hip.so)
export TMPDIR=/tmp
for rep in 1 2 3; do
for L in "$@"; do
  export IDE3D_HIP_LIB=$PWD/ide-3d_amd/$L/libide3d_hip.so
  echo -n "$L: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
