#!/bin/bash
# round 4: the ray-grid gather with static depth segments (hardware dispatch as the load balancer) before building a ticket scheme
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
GB=scripts/micro/bin/gather_bench
for s in 1 2 3 4 6 8; do
  echo "== IDE3D_GATHER_SEGS=$s"; IDE3D_GATHER_SEGS=$s GB_ITERS=400 timeout 200 $GB scripts/micro/bin/gather_coords.bin ide-3d_amd/lib/libide3d_hip.so | grep -E "avg|equal|differ" | head -4
done
