export TMPDIR=/tmp
GB=scripts/micro/bin/gather_bench; C=scripts/micro/bin/gather_coords.bin; L=ide-3d_amd
mkdir -p gpurun_out
( for f in 8; do echo "== IDE3D_GATHER_PC=$f"; IDE3D_GATHER_PC=$f GB_ITERS=400 timeout 200 $GB $C $L/lib/libide3d_hip.so $L/lib_pc_tprio/libide3d_hip.so; done
  for f in 8; do echo "== trace IDE3D_GATHER_PC=$f"; IDE3D_GATHER_PC=$f GB_ITERS=30 timeout 120 $GB $C $L/lib_pc_trace/libide3d_hip.so; done ) > gpurun_out/r3_gather3.log 2>&1
for f in 8; do IDE3D_GATHER_PC=$f timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "triplane or gather or sample_from" 2>&1 | tail -3; done > gpurun_out/r3_gather3_tests.log 2>&1
cat gpurun_out/r3_gather3_tests.log; cat gpurun_out/r3_gather3.log
