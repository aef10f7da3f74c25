"""`torch_utils.ops` call surface of the reference, backed by libide3d_hip.so on ROCm devices."""
