"""Minimal `dnnlib` for the render path: EasyDict + util (the reference's dnnlib/__init__.py re-exports these)."""

# Overlay package: modules this package does not carry (e.g. the reference's `dnnlib.seg_tools / dnnlib.camera`) resolve to the same-named package
# further down sys.path — put this tree in front of the reference checkout and its untouched scripts keep importing everything.
import pkgutil as _pkgutil
__path__ = _pkgutil.extend_path(__path__, __name__)
from .util import EasyDict, construct_class_by_name, get_obj_by_name  # noqa: F401
from . import util  # noqa: F401
