"""Render-path half of the reference `training` package (volumetric renderer, StyleGAN2 blocks, tri-plane generator)."""

# Overlay package: modules this package does not carry (e.g. the reference's `training.dataset`) resolve to the same-named package
# further down sys.path — put this tree in front of the reference checkout and its untouched scripts keep importing everything.
import pkgutil as _pkgutil
__path__ = _pkgutil.extend_path(__path__, __name__)
