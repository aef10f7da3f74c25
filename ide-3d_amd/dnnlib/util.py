"""`dnnlib.util` pieces the render path needs: EasyDict, by-name construction and the tri-plane /
grid feature samplers (reference dnnlib/util.py:46, :242-310, :561-617).

`sample_from_triplane` is the drop-in point of the HIP gather kernel (`csrc/triplane.hip`): float32
device tensors take the kernel, everything else (CPU, other dtypes) takes the PyTorch definition.
"""

import importlib
import sys
import types
from typing import Any, Tuple

import torch

from torch_utils import custom_ops
from torch_utils.ops import grid_sample_gradfix


class EasyDict(dict):
    """dict with attribute access."""

    def __getattr__(self, name: str) -> Any:
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = value

    def __delattr__(self, name: str) -> None:
        del self[name]


# ---- by-name object construction (reference util.py:242-310) -------------------------------------

def get_module_from_obj_name(obj_name: str) -> Tuple[types.ModuleType, str]:
    """Split 'pkg.mod.Obj.attr' into (imported module, 'Obj.attr'), trying the longest module path first."""
    parts = obj_name.split('.')
    last_err = None
    for i in range(len(parts), 0, -1):
        module_name, local_name = '.'.join(parts[:i]), '.'.join(parts[i:])
        try:
            module = importlib.import_module(module_name)
        except ImportError as e:
            if not str(e).startswith("No module named '" + module_name.split('.')[0]) and module_name in str(e):
                last_err = e
            continue
        try:
            get_obj_from_module(module, local_name)
            return module, local_name
        except AttributeError as e:
            last_err = e
    raise ImportError(f'cannot resolve "{obj_name}"') from last_err


def get_obj_from_module(module: types.ModuleType, obj_name: str) -> Any:
    obj = module
    if obj_name:
        for part in obj_name.split('.'):
            obj = getattr(obj, part)
    return obj


def get_obj_by_name(name: str) -> Any:
    module, local = get_module_from_obj_name(name)
    return get_obj_from_module(module, local)


def call_func_by_name(*args, func_name: str = None, **kwargs) -> Any:
    assert func_name is not None
    fn = get_obj_by_name(func_name)
    assert callable(fn)
    return fn(*args, **kwargs)


def construct_class_by_name(*args, class_name: str = None, **kwargs) -> Any:
    return call_func_by_name(*args, func_name=class_name, **kwargs)


# ---- feature samplers ---------------------------------------------------------------------------------

_triplane_plugin = None


def _triplane_init():
    global _triplane_plugin
    if _triplane_plugin is None:
        _triplane_plugin = custom_ops.get_plugin(module_name='triplane_plugin', sources=['triplane.hip'], headers=['triplane_tap.h'])
    return True


class _TriplaneSampleHip(torch.autograd.Function):
    """HIP tri-plane gather with gradients w.r.t. planes and coordinates."""

    @staticmethod
    def forward(ctx, coordinates, grid, ray_grid=None):
        if grid.stride(1) != 1:
            # NCHW planes: one channels_last copy so that every bilinear tap is a contiguous C*4-byte read.
            grid = grid.contiguous(memory_format=torch.channels_last)
        out = _triplane_plugin.sample(grid, coordinates, ray_grid=ray_grid)
        ctx.save_for_backward(coordinates, grid)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        coordinates, grid = ctx.saved_tensors
        grad_planes, grad_coords = _triplane_plugin.sample_backward(grad_out, grid, coordinates, ctx.needs_input_grad[0])
        return (grad_coords if ctx.needs_input_grad[0] else None), (grad_planes if ctx.needs_input_grad[1] else None), None


def sample_from_3dgrid(coordinates, grid):
    """Trilinear look-up: coordinates [B, M, 3], grid [1 or B, C, H, W, D] -> [B, M, C]
    (align_corners=True, zeros padding — reference util.py:561-577)."""
    coordinates = coordinates.float()
    grid = grid.float()
    b, m, d = coordinates.shape
    feats = torch.nn.functional.grid_sample(grid.expand(b, -1, -1, -1, -1), coordinates.reshape(b, 1, 1, -1, d),
                                            mode='bilinear', padding_mode='zeros', align_corners=True)
    n, c, h, w, dd = feats.shape
    return feats.permute(0, 4, 3, 2, 1).reshape(n, h * w * dd, c)


def sample_from_2dgrid(coordinates, grid):
    """Bilinear look-up: coordinates [B, M, 2], grid [B, C, H, W] -> [B*M, C] (reference util.py:603-617)."""
    b = grid.shape[0]
    feats = grid_sample_gradfix.grid_sample(grid, coordinates.reshape(b, -1, 1, 2))
    n, c, h, w = feats.shape
    return feats.permute(0, 3, 2, 1).reshape(n * h * w, c)


def _sample_from_triplane_ref(coordinates, grid):
    n, c3, h, w = grid.shape
    planes = grid.reshape(n, 3, c3 // 3, h, w)
    xy = sample_from_2dgrid(coordinates[..., [0, 1]], planes[:, 0])
    yz = sample_from_2dgrid(coordinates[..., [1, 2]], planes[:, 1])
    xz = sample_from_2dgrid(coordinates[..., [0, 2]], planes[:, 2])
    return xy + yz + xz


# ---- is a plain [B, M, 3] coordinate tensor a flattened ray grid? ------------------------------------------------------------------
# The LDS-staged gather kernel (csrc/triplane_tile.hip) tiles the samples as [rays_h, rays_w, steps]; its results are bit-equal to the
# flat kernel's for ANY coordinates and ANY factorisation of M (tiles whose footprint does not fit LDS read the planes directly), so the
# factorisation is purely a speed hint — and a reference caller does not pass one (volumetric_rendering.py:123-136 returns points that the
# generator reshapes to [B, R * S, 3] before `sample_from_triplane`).  The first call with a given M looks at the data once: the samples
# of one ray are collinear, so the ray length S is the index of the first point that leaves the line through points 0 and 1; the row
# length is where the first samples of consecutive rays stop advancing by one pixel step.  That costs one small device-to-host copy (a synchronisation), so the answer is cached per (M, device) and the
# look is skipped while a hipGraph is being captured (the flat kernel runs then, unless the answer is cached already).
_ray_grid_cache = {}


def _guess_ray_grid(coordinates):
    if coordinates.ndim != 3 or coordinates.shape[2] != 3 or coordinates.requires_grad:
        return None
    m = int(coordinates.shape[1])
    key = (m, coordinates.device.index)
    if key in _ray_grid_cache:
        return _ray_grid_cache[key]
    if m < 256 or torch.cuda.is_current_stream_capturing():
        return None
    head = coordinates[0, :min(m, 2048)].detach().double().cpu()          # (one synchronisation per new M)
    found = None
    d = head[1] - head[0]
    if float(d.norm()) > 0:
        rel = head - head[0]
        off = torch.linalg.cross(rel, d.expand_as(rel)).norm(dim=1)          # distance from the line through points 0 and 1, times |d|
        bad = (off > 1e-4 * float(d.norm()) * rel.norm(dim=1).clamp_min(float(d.norm()))).nonzero()
        if bad.numel():
            s = int(bad[0])
            if s >= 4 and s % 4 == 0 and m % s == 0:
                rays = m // s
                # row length: the first samples of consecutive rays advance by one pixel step along an image row and jump back at its end
                first = coordinates[0, ::s][:min(rays, 4096)].detach().double().cpu()
                step = (first[1:] - first[:-1]).norm(dim=1)
                jump = (step > 4 * float(step[0])).nonzero() if step.numel() and float(step[0]) > 0 else step.new_zeros(0)
                w = int(jump[0]) + 1 if jump.numel() else int(round(rays ** 0.5))
                if w >= 8 and rays % w == 0 and w % 8 == 0 and (rays // w) % 8 == 0:
                    found = (rays // w, w, s)
    if len(_ray_grid_cache) > 256:
        _ray_grid_cache.clear()
    _ray_grid_cache[key] = found
    return found


def sample_from_triplane(coordinates, grid, impl='cuda', ray_grid=None):
    """Sum of the xy / yz / xz plane look-ups: coordinates [B, M, 3], grid [B, 3*C, H, W] -> [B*M, C]
    (reference util.py:580-599; planes are square in every caller).

    `ray_grid=(rays_h, rays_w, steps)` (ours, optional): the M samples are a flattened [rays_h, rays_w, steps] ray grid, as
    produced by `transform_sampled_points`; lets the HIP library stage shared texels in LDS.  Results do not depend on it, and a
    caller need not pass it: without the hint the first call with a given M checks the data once (`_guess_ray_grid`);
    `ray_grid=False` forces the flat kernel."""
    assert impl in ['ref', 'cuda']
    use_hip = (impl == 'cuda' and grid.device.type == 'cuda' and grid.dtype == torch.float32
               and coordinates.dtype == torch.float32 and grid.shape[2] == grid.shape[3]
               and not grid_sample_gradfix.enabled)
    if use_hip and _triplane_init():
        if ray_grid is None:
            ray_grid = _guess_ray_grid(coordinates)          # the call exactly as the reference spells it (util.py:580): no hint needed
        elif ray_grid is False:
            ray_grid = None                                  # the flat kernel, whatever the coordinates look like (tests, roofline rows)
        return _TriplaneSampleHip.apply(coordinates, grid, ray_grid)
    return _sample_from_triplane_ref(coordinates, grid)


def layout_grid(img, grid_w=None, grid_h=1, float_to_uint8=True, chw_to_hwc=True, to_numpy=True):
    """Tile a batch [B, C, H, W] into one [grid_h*H, grid_w*W, C] uint8 image (reference util.py:632-646)."""
    b, c, h, w = img.shape
    if grid_w is None:
        grid_w = b // grid_h
    assert b == grid_w * grid_h
    if float_to_uint8:
        img = (img * 127.5 + 128).clamp(0, 255).to(torch.uint8)
    img = img.reshape(grid_h, grid_w, c, h, w).permute(2, 0, 3, 1, 4).reshape(c, grid_h * h, grid_w * w)
    if chw_to_hwc:
        img = img.permute(1, 2, 0)
    if to_numpy:
        img = img.cpu().numpy()
    return img


# ---- what the untouched driver scripts ask of `dnnlib.util` beyond the render path ---------------------------------------

def open_url(url: str, cache_dir: str = None, num_attempts: int = 10, verbose: bool = True, return_filename: bool = False,
             cache: bool = True) -> Any:
    """Binary file object of a network pickle (reference util.py:402): local paths and file:// URLs.  Anything that needs the
    network is handed to the reference's own `open_url` when its `dnnlib` sits behind this overlay on sys.path."""
    import re
    if not re.match('^[a-z]+://', url):
        return url if return_filename else open(url, 'rb')
    if url.startswith('file://'):
        import urllib.parse
        filename = urllib.parse.urlparse(url).path
        if re.match(r'^/[a-zA-Z]:', filename):
            filename = filename[1:]
        return filename if return_filename else open(filename, 'rb')
    ref = _reference_util()
    if ref is None:
        raise IOError(f'open_url: {url!r} needs the network; only local files and file:// URLs are handled by the overlay')
    return ref.open_url(url, cache_dir=cache_dir, num_attempts=num_attempts, verbose=verbose, return_filename=return_filename, cache=cache)


_ref_util = False          # False = not looked for yet; None = no reference checkout behind the overlay; else the loaded module
_ref_util_error = None     # the exception of a failed load (cached: the outcome must not depend on how often / in which order it is asked for)


def _reference_util():
    """The reference's own dnnlib/util.py, when a reference checkout sits behind this overlay on sys.path (the overlay package
    extends its __path__ over it), loaded under a private name; None otherwise.  A reference file that is present but cannot be
    imported (it imports `requests` etc. at module level) raises ImportError — every time, with the original cause chained."""
    global _ref_util, _ref_util_error
    if _ref_util_error is not None:
        raise ImportError(f"the reference's dnnlib/util.py behind the overlay could not be imported: {_ref_util_error!r}") from _ref_util_error
    if _ref_util is False:
        import os
        import importlib.util
        import dnnlib
        found = None
        here = os.path.dirname(os.path.abspath(__file__))
        for d in list(dnnlib.__path__):
            cand = os.path.join(d, 'util.py')
            if os.path.abspath(d) != here and os.path.isfile(cand):
                spec = importlib.util.spec_from_file_location('dnnlib._reference_util', cand)
                mod = importlib.util.module_from_spec(spec)
                try:
                    spec.loader.exec_module(mod)
                except Exception as e:      # noqa: BLE001 - whatever the foreign module raises while importing
                    _ref_util_error = e
                    raise ImportError(f"the reference's dnnlib/util.py ({cand}) could not be imported: {e!r}") from e
                found = mod
                break
        _ref_util = found
    return _ref_util


def __getattr__(name: str) -> Any:
    """Helpers this overlay does not re-state (Logger, format_time, make_cache_dir_path, ...) come from the reference's util."""
    if name.startswith('__'):
        raise AttributeError(name)
    try:
        ref = _reference_util()
    except ImportError as e:
        # module __getattr__ must answer AttributeError (hasattr / getattr-with-default rely on it); the cause stays attached
        raise AttributeError(f"module 'dnnlib.util' has no attribute {name!r} (and the reference's util.py behind the overlay failed to import)") from e
    if ref is not None and hasattr(ref, name):
        return getattr(ref, name)
    raise AttributeError(f"module 'dnnlib.util' has no attribute {name!r}")
