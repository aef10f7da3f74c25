#!/bin/bash
# round 3: how much of the gather's time are the planes that do not fit LDS?  Same kernel, three sets of cameras:
# bench (yaw -0.5, -0.15, 0.2, 0.5: 0.76 direct planes per chunk), frontal x4 (0.40), yaw +-0.5 x4 (0.96)
# coordinate files (git-ignored build products): GATHER_YAWS=0,0,0,0 GATHER_COORDS_NAME=gather_coords_frontal.bin python scripts/micro/make_gather_coords.py
#                                                GATHER_YAWS=0.5,0.5,-0.5,-0.5 GATHER_COORDS_NAME=gather_coords_yaw05.bin python scripts/micro/make_gather_coords.py
export TMPDIR=/tmp
cd /root/repo
GB=scripts/micro/bin/gather_bench; L=ide-3d_amd
for c in gather_coords.bin gather_coords_frontal.bin gather_coords_yaw05.bin; do
  echo "== $c"; GB_ITERS=400 timeout 200 $GB scripts/micro/bin/$c $L/lib/libide3d_hip.so | grep -E "tile +avg"
done
