"""Overlay of the reference's `inversion` package for the editing loop: `inversion.BiSeNet` / `inversion.resnet` are the MI355X modules of
training/face_parsing.py, so the reference's unchanged `dnnlib/seg_tools.py` (`from inversion.BiSeNet import BiSeNet`, :10) and Painter/run_UI.py
get the HIP convolutions; everything else (`inversion.networks`, ...) resolves to the reference's files further down sys.path."""

import pkgutil as _pkgutil
__path__ = _pkgutil.extend_path(__path__, __name__)
