"""Host-side mirror of the reference's `torch_utils` package for the IDE-3D render path (MI355X build)."""

# Overlay package: modules this package does not carry (e.g. the reference's `torch_utils.training_stats`) resolve to the same-named package
# further down sys.path — put this tree in front of the reference checkout and its untouched scripts keep importing everything.
import pkgutil as _pkgutil
__path__ = _pkgutil.extend_path(__path__, __name__)
