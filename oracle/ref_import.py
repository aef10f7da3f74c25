"""Import the REFERENCE's own Python modules from /root/reference (build container only; the GPU box has no copy).

The reference imports three modules that this image lacks: `cv2` (dnnlib/util.py:20) and the numpy-1 private
modules `numpy.lib.arraysetops` (inversion/networks.py:19) / `numpy.lib.function_base` (dnnlib/camera.py:5).
They are stubbed; nothing on the render path uses them.  TEST INFRASTRUCTURE ONLY.
"""

import os
import sys
import types

REFERENCE_ROOT = os.environ.get('IDE3D_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'torch_utils'))


def install():
    """Put the reference on sys.path (front) with the stubs in place.  Must run before any `torch_utils` import."""
    if not available():
        raise RuntimeError(f'reference tree not found at {REFERENCE_ROOT}')
    for mod in ('torch_utils', 'training', 'dnnlib', 'inversion'):
        if mod in sys.modules and not getattr(sys.modules[mod], '__file__', '').startswith(REFERENCE_ROOT):
            raise RuntimeError(f'{mod} is already imported from elsewhere; use a fresh interpreter')
    import numpy as np
    import numpy.lib
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    for name, attrs in (('arraysetops', dict(isin=np.isin)), ('function_base', dict(angle=np.angle, iterable=np.iterable))):
        m = types.ModuleType('numpy.lib.' + name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules['numpy.lib.' + name] = m
        setattr(numpy.lib, name, m)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
