"""Plugin loader — the MI355X counterpart of the reference's JIT builder.

The reference `get_plugin(module_name, sources, headers, source_dir, **build_kwargs)`
(torch_utils/custom_ops.py:59) compiles a pybind11 CUDA extension per op at first use and caches it
in a process-global dict (:57,68-69).  Here every op lives in one pre-built C-ABI shared library
(`ide-3d_amd/lib/libide3d_hip.so`, sources in `ide-3d_amd/csrc`, built by `hipcc --offload-arch=gfx950`);
`get_plugin` keeps the reference signature, ignores the CUDA-specific build arguments and returns an
object exposing the same functions the pybind module exposed (`plugin.bias_act(...)`, ...).

Set `IDE3D_HIP_LIB` to load the library from elsewhere.  A missing library is a hard error.
"""

import os
import subprocess

from . import hip_plugin

verbosity = 'brief'  # 'none', 'brief', 'full' (kept for API compatibility)

_cached_plugins = dict()


def build_library(verbose=False):
    """Compile libide3d_hip.so in-tree with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'csrc')
    cmd = ['make', '-C', csrc, f'-j{os.cpu_count() or 4}']
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError('building libide3d_hip.so failed:\n' + res.stdout[-4000:])
    return hip_plugin.lib_path()


def get_plugin(module_name, sources=None, headers=None, source_dir=None, **build_kwargs):
    assert verbosity in ['none', 'brief', 'full']
    if module_name in _cached_plugins:
        return _cached_plugins[module_name]
    if module_name not in hip_plugin.PLUGINS:
        raise RuntimeError(f'unknown plugin "{module_name}"; available: {sorted(hip_plugin.PLUGINS)}')
    if verbosity == 'full':
        print(f'Loading HIP plugin "{module_name}" from {hip_plugin.lib_path()} ...')
    hip_plugin.load()   # raises RuntimeError if the library is absent / unusable
    plugin = hip_plugin.PLUGINS[module_name]
    _cached_plugins[module_name] = plugin
    return plugin
