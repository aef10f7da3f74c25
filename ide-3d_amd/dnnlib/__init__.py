"""Minimal `dnnlib` for the render path: EasyDict + util (the reference's dnnlib/__init__.py re-exports these)."""
from .util import EasyDict, construct_class_by_name, get_obj_by_name  # noqa: F401
from . import util  # noqa: F401
