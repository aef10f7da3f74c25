"""`torch_utils.ops` call surface of the reference, backed by libide3d_hip.so on ROCm devices."""

# Overlay package: modules this package does not carry (e.g. the reference's `torch_utils.ops.*.cu sources`) resolve to the same-named package
# further down sys.path — put this tree in front of the reference checkout and its untouched scripts keep importing everything.
import pkgutil as _pkgutil
__path__ = _pkgutil.extend_path(__path__, __name__)
