export TMPDIR=/tmp
GB=scripts/micro/bin/gather_bench; C=scripts/micro/bin/gather_coords.bin; L=ide-3d_amd
mkdir -p gpurun_out
( echo "== old kernel"; IDE3D_GATHER_PC=0 GB_ITERS=400 timeout 120 $GB $C $L/lib/libide3d_hip.so
  echo "== pc variants"; GB_ITERS=400 timeout 300 $GB $C $L/lib/libide3d_hip.so $L/lib_pc_bprio/libide3d_hip.so $L/lib_pc_sprio/libide3d_hip.so $L/lib_pc_swap/libide3d_hip.so $L/lib/libide3d_hip.so
  echo "== trace"; GB_ITERS=30 timeout 120 $GB $C $L/lib_pc_trace/libide3d_hip.so ) > gpurun_out/r3_gather1.log 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "triplane or gather or sample_from" > gpurun_out/r3_gather1_tests.log 2>&1
tail -5 gpurun_out/r3_gather1_tests.log; cat gpurun_out/r3_gather1.log
