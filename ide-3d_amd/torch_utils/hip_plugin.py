"""ctypes binding of libide3d_hip.so (C ABI: include/ide3d_hip.h).

This is the host half of the drop-in boundary.  The reference loads one pybind11 module per op
through `torch_utils.custom_ops.get_plugin` (custom_ops.py:59) and calls `_plugin.<fn>(tensors...)`
(bias_act.py:150, upfirdn2d.py:242, filtered_lrelu.py:217,228).  Here the same call shapes are kept:
every `*Plugin` class below exposes functions with the reference plugin's argument order, validates
like the reference's TORCH_CHECKs, allocates outputs with torch (memory format preserved) and hands
raw device pointers + the current HIP stream to the C ABI.  PyTorch is only plumbing (device memory,
streams); all arithmetic happens in the hand-written HIP kernels.

There is deliberately NO fallback in this file: if the shared library is missing, was built for a
different arch, or a launch fails, a RuntimeError is raised.
"""

import ctypes
import math
import os
import threading
import weakref

import torch

_LIB_ENV = 'IDE3D_HIP_LIB'          # override path of libide3d_hip.so
_ABI_VERSION = 8
AMAX_SLOTS, AMAX_STRIDE = 32, 64     # = IDE3D_AMAX_SLOTS / _STRIDE (include/ide3d_hip.h): slot k of an image's `amax` row is element k * 64
AMAX_FLOATS = AMAX_SLOTS * AMAX_STRIDE

_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}

_lock = threading.Lock()
_lib = None

# number of successful C-ABI launches per entry point (tests assert the native path really ran)
CALLS = {}


def lib_path():
    env = os.environ.get(_LIB_ENV)
    if env:
        return env
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.join(os.path.dirname(here), 'lib', 'libide3d_hip.so')


class _UpfirdnParams(ctypes.Structure):
    _fields_ = [
        ('x', ctypes.c_void_p), ('f', ctypes.c_void_p), ('y', ctypes.c_void_p),
        ('dtype', ctypes.c_int32),
        ('n', ctypes.c_int32), ('c', ctypes.c_int32), ('in_h', ctypes.c_int32), ('in_w', ctypes.c_int32),
        ('out_h', ctypes.c_int32), ('out_w', ctypes.c_int32),
        ('x_stride', ctypes.c_int64 * 4), ('y_stride', ctypes.c_int64 * 4),
        ('f_h', ctypes.c_int32), ('f_w', ctypes.c_int32),
        ('f_stride', ctypes.c_int64 * 2),
        ('up_x', ctypes.c_int32), ('up_y', ctypes.c_int32), ('down_x', ctypes.c_int32), ('down_y', ctypes.c_int32),
        ('pad_x0', ctypes.c_int32), ('pad_y0', ctypes.c_int32),
        ('flip', ctypes.c_int32), ('gain', ctypes.c_float), ('x_row_floats', ctypes.c_int32),
    ]


def _row_floats_readable(x):
    """Elements readable from the start of the row of `x` that starts last in its storage (every other row has at least as many): what
    `ide3d_upfirdn2d_params.x_row_floats` promises.  0 (no promise) for anything but positive-stride rank-4 tensors."""
    if x.ndim != 4 or any(st <= 0 for st in x.stride()):
        return 0
    last_row = x.storage_offset() + sum((x.shape[i] - 1) * x.stride(i) for i in range(3))
    total = x.untyped_storage().nbytes() // x.element_size()
    return int(max(0, min(total - last_row, 2 ** 31 - 1)))


class _UpfirdnEpilogue(ctypes.Structure):
    _fields_ = [
        ('add', ctypes.c_void_p), ('add_stride', ctypes.c_int64 * 4),
        ('noise', ctypes.c_void_p), ('noise_strength', ctypes.c_float),
        ('bias', ctypes.c_void_p),
        ('fused_act', ctypes.c_int32), ('act', ctypes.c_int32),
        ('alpha', ctypes.c_float), ('act_gain', ctypes.c_float), ('clamp', ctypes.c_float),
        ('y_amax', ctypes.c_void_p),
    ]


class _FlreluParams(ctypes.Structure):
    _fields_ = [
        ('x', ctypes.c_void_p), ('y', ctypes.c_void_p), ('b', ctypes.c_void_p), ('s', ctypes.c_void_p),
        ('fu', ctypes.c_void_p), ('fd', ctypes.c_void_p),
        ('dtype', ctypes.c_int32),
        ('n', ctypes.c_int32), ('c', ctypes.c_int32), ('in_h', ctypes.c_int32), ('in_w', ctypes.c_int32),
        ('out_h', ctypes.c_int32), ('out_w', ctypes.c_int32),
        ('x_stride', ctypes.c_int64 * 4), ('y_stride', ctypes.c_int64 * 4),
        ('fu_w', ctypes.c_int32), ('fu_h', ctypes.c_int32), ('fd_w', ctypes.c_int32), ('fd_h', ctypes.c_int32),
        ('fu_stride', ctypes.c_int64 * 2), ('fd_stride', ctypes.c_int64 * 2),
        ('up', ctypes.c_int32), ('down', ctypes.c_int32),
        ('pad_x0', ctypes.c_int32), ('pad_y0', ctypes.c_int32),
        ('s_w_bytes', ctypes.c_int32), ('s_h', ctypes.c_int32),
        ('s_ofs_x', ctypes.c_int32), ('s_ofs_y', ctypes.c_int32),
        ('sw_limit', ctypes.c_int32), ('sign_mode', ctypes.c_int32),
        ('flip', ctypes.c_int32),
        ('gain', ctypes.c_float), ('slope', ctypes.c_float), ('clamp', ctypes.c_float),
    ]


class _Lattice(ctypes.Structure):
    _fields_ = [('n', ctypes.c_int32), ('voxel_size', ctypes.c_float), ('corner', ctypes.c_float * 3), ('scale', ctypes.c_float)]


class _RenderParams(ctypes.Structure):
    _fields_ = [
        ('rays_d_cam', ctypes.c_void_p), ('z_lin', ctypes.c_void_p), ('cam2world', ctypes.c_void_p),
        ('jitter', ctypes.c_void_p), ('sigma_noise', ctypes.c_void_p),
        ('tex_planes', ctypes.c_void_p), ('geo_planes', ctypes.c_void_p),
        ('tex_stride', ctypes.c_int64 * 4), ('geo_stride', ctypes.c_int64 * 4),
        ('geo_w0', ctypes.c_void_p), ('geo_b0', ctypes.c_void_p), ('geo_w1', ctypes.c_void_p), ('geo_b1', ctypes.c_void_p),
        ('tex_w0', ctypes.c_void_p), ('tex_b0', ctypes.c_void_p), ('tex_w1', ctypes.c_void_p), ('tex_b1', ctypes.c_void_p),
        ('n', ctypes.c_int32), ('rays_per_img', ctypes.c_int32), ('steps', ctypes.c_int32),
        ('C', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32),
        ('hidden', ctypes.c_int32), ('feat_ch', ctypes.c_int32), ('seg_ch', ctypes.c_int32),
        ('clamp_mode', ctypes.c_int32), ('last_back', ctypes.c_int32), ('white_back', ctypes.c_int32),
        ('max_depth', ctypes.c_float),
        ('out_feat', ctypes.c_void_p), ('out_depth', ctypes.c_void_p), ('out_wsum', ctypes.c_void_p),
    ]


class _ModconvParams(ctypes.Structure):
    _fields_ = [
        ('x', ctypes.c_void_p), ('w', ctypes.c_void_p), ('styles', ctypes.c_void_p), ('dcoefs', ctypes.c_void_p),
        ('noise', ctypes.c_void_p), ('bias', ctypes.c_void_p), ('y', ctypes.c_void_p),
        ('n', ctypes.c_int32), ('cin', ctypes.c_int32), ('cout', ctypes.c_int32),
        ('h', ctypes.c_int32), ('w_', ctypes.c_int32), ('k', ctypes.c_int32),
        ('noise_strength', ctypes.c_float),
        ('act', ctypes.c_int32), ('alpha', ctypes.c_float), ('gain', ctypes.c_float), ('clamp', ctypes.c_float),
        ('mode', ctypes.c_int32), ('weights_packed', ctypes.c_int32),
        ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_int64),
        ('w_batch_stride', ctypes.c_int64), ('arith', ctypes.c_int32),
        ('x_amax', ctypes.c_void_p), ('y_amax', ctypes.c_void_p), ('y_pitch', ctypes.c_int32),
    ]


class _ModconvPlanInfo(ctypes.Structure):
    """ide3d_modconv_plan_info (include/ide3d_hip.h): what ide3d_modconv2d would launch."""
    _fields_ = [
        ('kind', ctypes.c_int32), ('tile_h', ctypes.c_int32), ('tile_w', ctypes.c_int32), ('images_per_tile', ctypes.c_int32),
        ('rows', ctypes.c_int32), ('waves', ctypes.c_int32), ('parts', ctypes.c_int32), ('f16', ctypes.c_int32),
        ('split_k', ctypes.c_int32), ('strip', ctypes.c_int32), ('transposed_all_class', ctypes.c_int32), ('reserved', ctypes.c_int32),
        ('workgroups', ctypes.c_int64),
    ]


PLAN_KINDS = ('fp32', 'split', 'split_teams', 'head_split', 'head_resident', 'head_small')     # IDE3D_PLAN_*

STYLE_BATCH_MAX = 24        # IDE3D_STYLE_BATCH_MAX


class _StyleJob(ctypes.Structure):
    _fields_ = [
        ('w', ctypes.c_void_p), ('w_stride', ctypes.c_int64),
        ('affine_w', ctypes.c_void_p), ('affine_b', ctypes.c_void_p),
        ('wsq_t', ctypes.c_void_p),
        ('cin', ctypes.c_int32), ('cout', ctypes.c_int32),
        ('affine_gain', ctypes.c_float), ('bias_gain', ctypes.c_float),
        ('styles', ctypes.c_void_p), ('dcoefs', ctypes.c_void_p),
    ]


class _FoldJob(ctypes.Structure):
    _fields_ = [
        ('w', ctypes.c_void_p), ('w_stride', ctypes.c_int64),
        ('cin', ctypes.c_int32), ('affine_gain', ctypes.c_float),
        ('a0', ctypes.c_void_p), ('b0', ctypes.c_void_p), ('w0', ctypes.c_void_p), ('cout0', ctypes.c_int32), ('gain0', ctypes.c_float),
        ('a1', ctypes.c_void_p), ('b1', ctypes.c_void_p), ('w1', ctypes.c_void_p), ('cout1', ctypes.c_int32), ('gain1', ctypes.c_float),
        ('out', ctypes.c_void_p),
    ]


LOWRES_MAX_LAYERS, LOWRES_MAX_HEADS = 8, 4          # IDE3D_LOWRES_MAX_*


class _LowresLayer(ctypes.Structure):
    _fields_ = [('weight', ctypes.c_void_p), ('styles', ctypes.c_void_p), ('dcoefs', ctypes.c_void_p), ('noise', ctypes.c_void_p), ('bias', ctypes.c_void_p),
                ('act_gain', ctypes.c_float), ('clamp', ctypes.c_float), ('up', ctypes.c_int32), ('head', ctypes.c_int32),
                ('weights_packed', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class _LowresHead(ctypes.Structure):
    _fields_ = [('w', ctypes.c_void_p), ('bias', ctypes.c_void_p), ('skip', ctypes.c_void_p), ('clamp', ctypes.c_float), ('O', ctypes.c_int32)]


class _LowresParams(ctypes.Structure):
    _fields_ = [('x0', ctypes.c_void_p), ('x0_batch_stride', ctypes.c_int64), ('fir', ctypes.c_void_p), ('x_out', ctypes.c_void_p),
                ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_int64),
                ('n', ctypes.c_int32), ('C', ctypes.c_int32), ('res0', ctypes.c_int32), ('nlayers', ctypes.c_int32), ('nheads', ctypes.c_int32),
                ('arith', ctypes.c_int32), ('persistent', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('layers', _LowresLayer * LOWRES_MAX_LAYERS), ('heads', _LowresHead * LOWRES_MAX_HEADS)]


class _MappingParams(ctypes.Structure):
    _fields_ = [
        ('z', ctypes.c_void_p), ('c', ctypes.c_void_p), ('embed_w', ctypes.c_void_p), ('embed_b', ctypes.c_void_p),
        ('fc_w', ctypes.c_void_p * 16), ('fc_b', ctypes.c_void_p * 16), ('fc_out', ctypes.c_int32 * 16),
        ('w_avg', ctypes.c_void_p), ('ws', ctypes.c_void_p),
        ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_int64),
        ('n', ctypes.c_int32), ('z_dim', ctypes.c_int32), ('c_dim', ctypes.c_int32), ('embed', ctypes.c_int32),
        ('layers', ctypes.c_int32), ('num_ws', ctypes.c_int32),
        ('embed_weight_gain', ctypes.c_float), ('embed_bias_gain', ctypes.c_float), ('lr_multiplier', ctypes.c_float),
        ('alpha', ctypes.c_float), ('act_gain', ctypes.c_float), ('truncation_psi', ctypes.c_float),
        ('truncation_cutoff', ctypes.c_int32),
    ]


def _hip_runtimes_mapped():
    """Paths of every libamdhip64 mapped into this process (there must be exactly one)."""
    paths = set()
    try:
        with open('/proc/self/maps') as f:
            for line in f:
                if 'libamdhip64' in line:
                    paths.add(line.split()[-1])
    except OSError:
        pass
    return paths


def load():
    """dlopen libide3d_hip.so once and declare the prototypes.  Raises RuntimeError if unusable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.isfile(path):
            raise RuntimeError(
                f'libide3d_hip.so not found at {path}. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'or `make -C ide-3d_amd/csrc`, or point ${_LIB_ENV} at it. There is no CPU/eager fallback for CUDA tensors.')
        # torch ships its own libamdhip64.so (soname libamdhip64.so.7); make sure it is the one already mapped
        # so that our kernels share torch's HIP runtime, streams and allocations.
        torch.cuda.is_available()
        try:
            lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        except OSError as e:
            raise RuntimeError(f'failed to load {path}: {e}') from e
        rts = _hip_runtimes_mapped()
        if len(rts) > 1:
            raise RuntimeError(f'two HIP runtimes mapped in one process ({sorted(rts)}); import torch before loading the plugin')
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
        lib.ide3d_last_error.restype = ctypes.c_char_p
        lib.ide3d_last_error.argtypes = []
        lib.ide3d_abi_version.restype = ctypes.c_int
        lib.ide3d_build_arch.restype = ctypes.c_char_p
        if lib.ide3d_abi_version() != _ABI_VERSION:
            raise RuntimeError(f'{path}: ABI version {lib.ide3d_abi_version()} != expected {_ABI_VERSION}; rebuild')
        # a library built with timing-only experiment knobs (wrong results by design) or without exclusive residency must never be
        # mistaken for the product: it loads only when the experimenter says so
        lib.ide3d_build_flags.restype = ctypes.c_char_p
        lib.ide3d_build_flags.argtypes = []
        flags = lib.ide3d_build_flags().decode('ascii', 'replace').split()
        unsafe = [f for f in flags if f.startswith('!')]
        if unsafe and os.environ.get('IDE3D_ALLOW_EXPERIMENT_BUILD') != '1':
            raise RuntimeError(f'{path} was built with experiment knobs that change results or drop a safety property ({" ".join(unsafe)}); '
                               'rebuild without EXTRA=..., or set IDE3D_ALLOW_EXPERIMENT_BUILD=1 for timing experiments')
        protos = {
            'ide3d_bias_act': [vp, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32, f32, f32, i64, i64, i64, vp],
            'ide3d_upfirdn2d': [ctypes.POINTER(_UpfirdnParams), vp],
            'ide3d_upfirdn2d_ex': [ctypes.POINTER(_UpfirdnParams), ctypes.POINTER(_UpfirdnEpilogue), vp],
            'ide3d_filtered_lrelu': [ctypes.POINTER(_FlreluParams), vp],
            'ide3d_filtered_lrelu_act': [vp, vp, ctypes.c_int, i32, i32, i32, i32, ctypes.POINTER(i64 * 4),
                                         i32, i32, i32, i32, f32, f32, f32, ctypes.c_int, vp],
            'ide3d_triplane_sample': [vp, ctypes.POINTER(i64 * 4), i32, i32, i32, i32, vp, i64, vp, vp],
            'ide3d_triplane_sample_rays': [vp, ctypes.POINTER(i64 * 4), i32, i32, i32, i32, vp, i64, vp, i32, i32, i32, vp],
            'ide3d_triplane_taps': [i32, i32, vp, i64, vp, vp],
            'ide3d_triplane_sample_backward': [vp, vp, ctypes.POINTER(i64 * 4), i32, i32, i32, i32, vp, i64,
                                               vp, ctypes.POINTER(i64 * 4), vp, vp],
            'ide3d_composite': [vp, vp, vp, vp, i64, i32, i32, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32,
                                ctypes.c_int, vp, vp, vp, vp],
            'ide3d_sample_pdf': [vp, vp, vp, i64, i64, i32, i32, f32, vp, vp],
            'ide3d_render_rays': [ctypes.POINTER(_RenderParams), vp],
            'ide3d_sample_voxel': [ctypes.POINTER(_RenderParams), vp, i64, vp, vp, ctypes.c_int, vp],
            'ide3d_lattice_points': [ctypes.POINTER(_Lattice), i64, i64, vp, vp],
            'ide3d_density_lattice': [ctypes.POINTER(_RenderParams), ctypes.POINTER(_Lattice), i64, i64, vp, vp],
            'ide3d_modconv2d': [ctypes.POINTER(_ModconvParams), vp],
            'ide3d_modconv_plan': [ctypes.POINTER(_ModconvParams), ctypes.POINTER(_ModconvPlanInfo)],
            'ide3d_modconv_workspace_bytes': [i32, i32, i32, i32, i32, i32, i32, i32],
            'ide3d_set_conv_arithmetic': [i32],
            'ide3d_get_conv_arithmetic': [],
            'ide3d_frame_u8': [vp, vp, vp, i32, i32, i32, i32, vp, vp],
            'ide3d_sphere_points': [vp, vp, i32, ctypes.c_float, i32, vp, vp, vp],
            'ide3d_cam2world': [vp, vp, vp, i32, i32, vp, vp],
            'ide3d_style_demod': [vp, i64, vp, vp, vp, i32, i32, i32, i32, f32, f32, vp, vp, vp],
            'ide3d_fold_heads': [vp, i64, i32, i32, i32, f32, vp, vp, vp, i32, f32, vp, vp, vp, i32, f32, vp, vp],
            'ide3d_style_demod_batch': [ctypes.POINTER(_StyleJob), i32, i32, i32, vp],
            'ide3d_fold_heads_batch': [ctypes.POINTER(_FoldJob), i32, i32, i32, vp],
            'ide3d_mapping': [ctypes.POINTER(_MappingParams), vp],
            'ide3d_mapping_workspace_bytes': [],
            'ide3d_mapping_supported': [],
            'ide3d_skip_upsample_add_cl': [vp, ctypes.POINTER(i64 * 4), vp, ctypes.POINTER(i64 * 4), i32, i32, i32, i32, vp, vp],
            'ide3d_bilinear_up2_split': [vp, i32, i32, i32, i32, ctypes.POINTER(vp * 3), ctypes.POINTER(i32 * 3), ctypes.POINTER(i32 * 3), ctypes.POINTER(ctypes.c_int64 * 3), vp],
            'ide3d_lowres_layers_supported': [i32, i32, i32, ctypes.POINTER(i32), i32, i32],
            'ide3d_lowres_workspace_bytes': [ctypes.POINTER(_LowresParams)],
            'ide3d_lowres_group': [ctypes.POINTER(_LowresParams), vp],
        }
        for name, argtypes in protos.items():
            fn = getattr(lib, name)          # AttributeError here = header / library mismatch
            fn.restype = ctypes.c_int64 if name.endswith('_bytes') else ctypes.c_int
            fn.argtypes = argtypes
        _lib = lib
        return _lib


EXPORTED_SYMBOLS = (
    'ide3d_last_error', 'ide3d_abi_version', 'ide3d_build_arch', 'ide3d_build_flags', 'ide3d_exclusive_violations', 'ide3d_exclusive_violation_text', 'ide3d_bias_act', 'ide3d_upfirdn2d', 'ide3d_upfirdn2d_ex',
    'ide3d_filtered_lrelu', 'ide3d_filtered_lrelu_act', 'ide3d_triplane_sample', 'ide3d_triplane_sample_rays', 'ide3d_triplane_taps',
    'ide3d_triplane_sample_backward', 'ide3d_composite', 'ide3d_sample_pdf', 'ide3d_render_rays', 'ide3d_sample_voxel',
    'ide3d_lattice_points', 'ide3d_density_lattice',
    'ide3d_modconv2d', 'ide3d_modconv_workspace_bytes', 'ide3d_modconv_plan', 'ide3d_set_conv_arithmetic', 'ide3d_get_conv_arithmetic', 'ide3d_frame_u8', 'ide3d_sphere_points', 'ide3d_cam2world', 'ide3d_style_demod', 'ide3d_fold_heads',
    'ide3d_style_demod_batch', 'ide3d_fold_heads_batch',
    'ide3d_skip_upsample_add_cl', 'ide3d_bilinear_up2_split', 'ide3d_mapping', 'ide3d_mapping_workspace_bytes', 'ide3d_mapping_supported',
    'ide3d_lowres_layers_supported', 'ide3d_lowres_workspace_bytes', 'ide3d_lowres_group',
)


def exclusive_violations():
    """(count, text): kernels with an LDS-fed bf16 / fp16 matrix loop that would NOT be alone on their CU on this device (their launches are
    refused with a RuntimeError); (0, '') everywhere the library is supported.  See ide3d_exclusive_violations (include/ide3d_hip.h)."""
    lib = load()
    lib.ide3d_exclusive_violations.restype = ctypes.c_int
    lib.ide3d_exclusive_violation_text.restype = ctypes.c_char_p
    return int(lib.ide3d_exclusive_violations()), lib.ide3d_exclusive_violation_text().decode('utf-8', 'replace')


def _check(rc, what):
    if rc == 0:
        CALLS[what] = CALLS.get(what, 0) + 1
    if rc != 0:
        msg = load().ide3d_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'{what} failed (code {rc}): {msg}')


# Host cost per launch matters for the drop-in loop shape (batch 1, eager: ~75 launches of 10-40 us of GPU work each; bench.py
# `dropin_eager_b1`): the raw stream handle comes from torch's C entry point (no `Stream` object), and the device guard is a no-op when the
# tensor's device is already current (the common single-GPU case).
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream_handle(device):
    if _raw_stream is not None:
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


def _stream(t):
    return ctypes.c_void_p(_stream_handle(t.device))


# ---- streams of our own ----------------------------------------------------------------------------------------------------------
# `torch.cuda.Stream()` hands out one of 32 pooled handles per device, round-robin: the 33rd stream an application (or this library) asks
# for IS the first one again.  A hipGraph capture whose capture stream happens to be the style side stream, or a stream the caller renders
# on, is malformed (round 5: a host-side segfault in hipGraphLaunch at the 33rd capture of a process, scripts/micro/r5_graph_churn.py).
# The streams this library forks work to - the capture stream of training/graph_cache.py and triplane.GraphedRenderer, the style
# prefetch side stream - are therefore created from the HIP runtime directly, once per (device, purpose), and never collide with a pooled one.
_private_streams = {}


def _hip_runtime():
    paths = _hip_runtimes_mapped()
    if len(paths) != 1:
        raise RuntimeError(f'expected exactly one HIP runtime mapped in this process, found {sorted(paths)}')
    rt = ctypes.CDLL(next(iter(paths)))          # the path that is already mapped: the same instance torch uses
    rt.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    rt.hipStreamCreateWithFlags.restype = ctypes.c_int
    return rt


def private_stream(device, purpose):
    """A `torch.cuda.ExternalStream` around a HIP stream created for (device, purpose) and kept for the life of the process."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, purpose)
    st = _private_streams.get(key)
    if st is None:
        with _lock:
            st = _private_streams.get(key)
            if st is None:
                torch.cuda.init()
                h = ctypes.c_void_p()
                with torch.cuda.device(idx):
                    rc = _hip_runtime().hipStreamCreateWithFlags(ctypes.byref(h), 1)          # hipStreamNonBlocking, like torch's pooled streams
                if rc != 0 or not h.value:
                    raise RuntimeError(f'hipStreamCreateWithFlags failed ({rc})')
                st = _private_streams[key] = torch.cuda.ExternalStream(h.value, device=torch.device('cuda', idx))
    return st


capture_lock = threading.RLock()          # one hipGraph capture at a time per process (they share the capture stream of their device)


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _dev_guard(device):
    """`with _dev_guard(t.device):` does what `with torch.cuda.device(t.device):` does, minus the context switch when that device is current."""
    if device is None:
        return _NO_GUARD
    idx = device.index if isinstance(device, torch.device) else device
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(idx)


# ---- who owns a launch's scratch memory ------------------------------------------------------------------------------------------
# Split-K partials, packed weights and the mapping kernel's barrier counter live in workspaces that belong to ONE launch at a time.
# Eager callers are told apart by their stream (two streams never share a workspace).  A stream HANDLE is not an identity, though:
# PyTorch hands pooled handles out round-robin and a hipGraph is replayed on whatever stream is current, so a captured graph owns its
# workspaces through `workspace_scope(owner)` instead: everything launched inside the scope is keyed by the owner object (one
# `GraphedRenderer`), and the entries are dropped when the owner is garbage-collected.  Two graphs therefore never share scratch
# memory with each other or with eager callers; replays of ONE graph must be serialised (they share its static buffers anyway).
_scope = threading.local()
_scope_finalizers = {}


def _drop_owner(domain):
    _scope_finalizers.pop(domain, None)
    for cache in (ModconvPlugin._ws, MappingPlugin._ws, LowresPlugin._ws):
        for k in [k for k in cache if domain in k]:
            del cache[k]


class workspace_scope:
    """`with workspace_scope(owner): ...` — launches inside use workspaces owned by `owner` (any weak-referenceable object)."""

    def __init__(self, owner):
        self.domain = ('owner', id(owner))
        if self.domain not in _scope_finalizers:
            _scope_finalizers[self.domain] = weakref.finalize(owner, _drop_owner, self.domain)

    def __enter__(self):
        self.prev = getattr(_scope, 'domain', None)
        _scope.domain = self.domain
        return self

    def __exit__(self, *exc):
        _scope.domain = self.prev
        return False


def _ws_domain(device):
    """Key component that separates workspaces of concurrent launch sequences: the owner of the active scope, else the current stream."""
    dom = getattr(_scope, 'domain', None)
    return dom if dom is not None else ('stream', _stream_handle(device))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if (t is not None and t.numel() > 0) else ctypes.c_void_p(0)


def _i64x4(vals):
    return (ctypes.c_int64 * 4)(*[int(v) for v in vals])


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _dense(t):
    return t.is_contiguous() or (t.ndim == 4 and t.is_contiguous(memory_format=torch.channels_last))


# ------------------------------------------------------------------------------------------------
# bias_act_plugin
# ------------------------------------------------------------------------------------------------

class BiasActPlugin:
    """`bias_act_plugin` of the reference (bias_act.cpp:32-97)."""

    @staticmethod
    def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
        _require(x.is_cuda, 'x must reside on CUDA device')
        for name, t in (('b', b), ('xref', xref), ('yref', yref), ('dy', dy)):
            if t.numel():
                _require(t.device == x.device, f'{name} must reside on the same device as x')
                _require(t.dtype == x.dtype, f'{name} must have the same dtype as x')
        _require(x.dtype in _DTYPE_CODE, 'x must be float16, bfloat16, float32 or float64')
        _require(b.numel() == 0 or (b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]),
                 'b must be a vector matching dimension `dim` of x')
        _require(b.numel() == 0 or b.is_contiguous(), 'b must be contiguous')
        for name, t in (('xref', xref), ('yref', yref), ('dy', dy)):
            _require(t.numel() == 0 or (t.shape == x.shape and t.stride() == x.stride()),
                     f'{name} must have the same shape and layout as x')
        _require(grad >= 0, 'grad must be non-negative')
        _require(_dense(x) or x.numel() == 0 or x.is_non_overlapping_and_dense(), 'x must be non-overlapping and dense')
        y = torch.empty_like(x)
        _require(y.stride() == x.stride(), 'internal: output layout differs from input layout')
        if x.numel() == 0:
            return y
        step_b = x.stride(dim) if b.numel() else 1
        with _dev_guard(x.device):
            rc = load().ide3d_bias_act(_ptr(x), _ptr(b), _ptr(xref), _ptr(yref), _ptr(dy), _ptr(y),
                                       _DTYPE_CODE[x.dtype], int(grad), int(act), float(alpha), float(gain), float(clamp),
                                       x.numel(), max(b.numel(), 1), int(step_b), _stream(x))
        _check(rc, 'bias_act')
        return y


# ------------------------------------------------------------------------------------------------
# upfirdn2d_plugin
# ------------------------------------------------------------------------------------------------

class Upfirdn2dPlugin:
    """`upfirdn2d_plugin` of the reference (upfirdn2d.cpp:16-105)."""

    @staticmethod
    def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
        return Upfirdn2dPlugin.upfirdn2d_ex(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)

    @staticmethod
    def upfirdn2d_ex(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain,
                     add=None, noise=None, noise_strength=1.0, bias=None, act=None, alpha=0.0, act_gain=1.0, clamp=-1.0, y_amax=None):
        """upfirdn2d with the optional fused epilogue  y = bias_act(FIR(x) + add + noise * noise_strength)
        (`act` None = no bias/activation stage; 1 linear, 3 lrelu)."""
        _require(x.is_cuda, 'x must reside on CUDA device')
        _require(f.device == x.device, 'f must reside on the same device as x')
        _require(f.dtype == torch.float32, 'f must be float32')
        _require(x.dtype in _DTYPE_CODE, 'x must be float16, bfloat16, float32 or float64')
        _require(x.numel() > 0, 'x has zero size')
        _require(f.numel() > 0, 'f has zero size')
        _require(x.ndim == 4, 'x must be rank 4')
        _require(f.ndim == 2, 'f must be rank 2')
        _require(f.shape[0] >= 1 and f.shape[1] >= 1, 'f must be at least 1x1')
        _require(upx >= 1 and upy >= 1, 'upsampling factor must be at least 1')
        _require(downx >= 1 and downy >= 1, 'downsampling factor must be at least 1')
        n, c, ih, iw = x.shape
        ow = (iw * upx + padx0 + padx1 - f.shape[1] + downx) // downx
        oh = (ih * upy + pady0 + pady1 - f.shape[0] + downy) // downy
        _require(ow >= 1 and oh >= 1, 'output must be at least 1x1')
        cl = x.ndim == 4 and x.stride(1) == 1 and c > 1
        y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last if cl else torch.contiguous_format)
        p = _UpfirdnParams()
        p.x, p.f, p.y = x.data_ptr(), f.data_ptr(), y.data_ptr()
        p.dtype = _DTYPE_CODE[x.dtype]
        p.n, p.c, p.in_h, p.in_w, p.out_h, p.out_w = n, c, ih, iw, oh, ow
        p.x_stride = _i64x4(x.stride())
        p.y_stride = _i64x4(y.stride())
        p.f_h, p.f_w = f.shape
        p.f_stride = (ctypes.c_int64 * 2)(f.stride(0), f.stride(1))
        p.up_x, p.up_y, p.down_x, p.down_y = upx, upy, downx, downy
        p.pad_x0, p.pad_y0 = padx0, pady0
        p.flip, p.gain = int(bool(flip)), float(gain)
        p.x_row_floats = _row_floats_readable(x)
        plain = add is None and noise is None and act is None and y_amax is None
        with _dev_guard(x.device):
            if plain:
                rc = load().ide3d_upfirdn2d(ctypes.byref(p), _stream(x))
            else:
                ep = _UpfirdnEpilogue()
                keep = []
                if y_amax is not None:     # [n, AMAX_FLOATS] float32, zeroed by the caller: row max = max |y| per image (f16x3 scale of the consumer)
                    _require(y_amax.is_cuda and y_amax.device == x.device and y_amax.dtype == torch.float32 and tuple(y_amax.shape) == (n, AMAX_FLOATS)
                             and y_amax.is_contiguous() and x.dtype == torch.float32, 'y_amax must be a contiguous float32 [n, AMAX_FLOATS] tensor on the device of a float32 x')
                    ep.y_amax = y_amax.data_ptr()
                if add is not None:
                    _require(add.shape == y.shape and add.dtype == x.dtype and add.device == x.device, 'add must match the output')
                    ep.add, ep.add_stride = add.data_ptr(), _i64x4(add.stride())
                if noise is not None:
                    noise = noise.to(torch.float32).contiguous(); keep.append(noise)
                    _require(tuple(noise.shape[-2:]) == (oh, ow) and noise.numel() == oh * ow, 'noise must be [out_h, out_w]')
                    ep.noise, ep.noise_strength = noise.data_ptr(), float(noise_strength)
                if act is not None:
                    ep.fused_act, ep.act = 1, int(act)
                    ep.alpha, ep.act_gain, ep.clamp = float(alpha), float(act_gain), float(clamp)
                    if bias is not None:
                        bias = bias.to(x.dtype).contiguous(); keep.append(bias)
                        _require(bias.numel() == c, 'bias must have one entry per channel')
                        ep.bias = bias.data_ptr()
                rc = load().ide3d_upfirdn2d_ex(ctypes.byref(p), ctypes.byref(ep), _stream(x))
        _check(rc, 'upfirdn2d')
        return y


# ------------------------------------------------------------------------------------------------
# filtered_lrelu_plugin
# ------------------------------------------------------------------------------------------------

class FilteredLReluPlugin:
    """`filtered_lrelu_plugin` of the reference (filtered_lrelu.cpp:16-298)."""

    @staticmethod
    def filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp, flip_filters, write_signs):
        _require(x.is_cuda, 'x must reside on CUDA device')
        _require(fu.device == x.device and fd.device == x.device and b.device == x.device,
                 'all input tensors must reside on the same device')
        _require(fu.dtype == torch.float32 and fd.dtype == torch.float32, 'fu and fd must be float32')
        _require(b.dtype == x.dtype, 'x and b must have the same dtype')
        _require(x.dtype in (torch.float16, torch.float32, torch.bfloat16), 'x and b must be float16, bfloat16 or float32')
        _require(x.ndim == 4, 'x must be rank 4')
        _require(x.numel() > 0, 'x is empty')
        _require(fu.ndim in (1, 2) and fd.ndim in (1, 2), 'fu and fd must be rank 1 or 2')
        _require(fu.numel() > 0, 'fu is empty')
        _require(fd.numel() > 0, 'fd is empty')
        _require(b.ndim == 1 and b.shape[0] == x.shape[1], 'b must be a vector with the same number of channels as x')
        _require(up >= 1 and down >= 1, 'up and down must be at least 1')
        n, c, xh, xw = x.shape
        fut_w, fut_h = fu.shape[-1] - 1, fu.shape[0] - 1
        fdt_w, fdt_h = fd.shape[-1] - 1, fd.shape[0] - 1
        cw = xw * up + (px0 + px1) - fut_w
        ch = xh * up + (py0 + py1) - fut_h
        _require(cw > fdt_w and ch > fdt_h, 'upsampled buffer must be at least the size of downsampling filter')
        yw = (cw - fdt_w + (down - 1)) // down
        yh = (ch - fdt_h + (down - 1)) // down
        _require(yw > 0 and yh > 0, 'output must be at least 1x1')
        read_signs = si.numel() > 0
        so = torch.empty([0], dtype=torch.uint8, device=x.device)
        s = si
        sw_active = 0
        if write_signs:
            sw_active = yw * down - (down - 1) + fdt_w
            sh = yh * down - (down - 1) + fdt_h
            sw = (sw_active + 15) & ~15
            s = so = torch.empty([n, c, sh, sw >> 2], dtype=torch.uint8, device=x.device)
        elif read_signs:
            sw_active = s.shape[3] << 2
        if read_signs or write_signs:
            _require(s.is_contiguous(), 'signs must be contiguous')
            _require(s.dtype == torch.uint8, 'signs must be uint8')
            _require(s.device == x.device, 'signs must reside on the same device as x')
            _require(s.ndim == 4, 'signs must be rank 4')
            _require(s.shape[0] == n and s.shape[1] == c, 'signs must have same batch & channels as x')
        cl = x.stride(1) == 1 and c > 1
        y = torch.empty([n, c, yh, yw], dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last if cl else torch.contiguous_format)
        p = _FlreluParams()
        p.x, p.y, p.b = x.data_ptr(), y.data_ptr(), b.contiguous().data_ptr()
        p.s = s.data_ptr() if (read_signs or write_signs) else 0
        p.fu, p.fd = fu.data_ptr(), fd.data_ptr()
        p.dtype = _DTYPE_CODE[x.dtype]
        p.n, p.c, p.in_h, p.in_w, p.out_h, p.out_w = n, c, xh, xw, yh, yw
        p.x_stride, p.y_stride = _i64x4(x.stride()), _i64x4(y.stride())
        p.fu_w, p.fu_h = fu.shape[-1], (fu.shape[0] if fu.ndim == 2 else 0)
        p.fd_w, p.fd_h = fd.shape[-1], (fd.shape[0] if fd.ndim == 2 else 0)
        p.fu_stride = (ctypes.c_int64 * 2)(fu.stride(0) if fu.ndim == 2 else 0, fu.stride(-1))
        p.fd_stride = (ctypes.c_int64 * 2)(fd.stride(0) if fd.ndim == 2 else 0, fd.stride(-1))
        p.up, p.down, p.pad_x0, p.pad_y0 = up, down, px0, py0
        p.s_w_bytes, p.s_h = (s.shape[3], s.shape[2]) if (read_signs or write_signs) else (0, 0)
        p.s_ofs_x, p.s_ofs_y = sx, sy
        p.sw_limit = (sw_active + 3) >> 2
        p.sign_mode = 1 if write_signs else (2 if read_signs else 0)
        p.flip = int(bool(flip_filters))
        p.gain, p.slope, p.clamp = float(gain), float(slope), float(min(clamp, 3.0e38))
        with _dev_guard(x.device):
            rc = load().ide3d_filtered_lrelu(ctypes.byref(p), _stream(x))
        if rc == -2:     # IDE3D_ENOKERNEL: same contract as the reference's return_code = -1
            return torch.empty([0], device=x.device), torch.empty([0], device=x.device), -1
        _check(rc, 'filtered_lrelu')
        return y, so, 0

    @staticmethod
    def filtered_lrelu_act_(x, si, sx, sy, gain, slope, clamp, write_signs):
        _require(x.is_cuda, 'x must reside on CUDA device')
        _require(x.ndim == 4, 'x must be rank 4')
        _require(x.numel() > 0, 'x is empty')
        _require(x.dtype in _DTYPE_CODE, 'x must be float16, bfloat16, float32 or float64')
        n, c, h, w = x.shape
        read_signs = si.numel() > 0
        so = torch.empty([0], dtype=torch.uint8, device=x.device)
        s = si
        if write_signs:
            sw = (w + 15) & ~15
            s = so = torch.empty([n, c, h, sw >> 2], dtype=torch.uint8, device=x.device)
        if read_signs or write_signs:
            _require(s.is_contiguous(), 'signs must be contiguous')
            _require(s.dtype == torch.uint8, 'signs must be uint8')
            _require(s.device == x.device, 'signs must reside on the same device as x')
            _require(s.ndim == 4, 'signs must be rank 4')
            _require(s.shape[0] == n and s.shape[1] == c, 'signs must have same batch & channels as x')
        s_w, s_h = ((s.shape[3] << 2), s.shape[2]) if (read_signs or write_signs) else (0, 0)
        with _dev_guard(x.device):
            rc = load().ide3d_filtered_lrelu_act(
                _ptr(x), _ptr(s) if (read_signs or write_signs) else ctypes.c_void_p(0), _DTYPE_CODE[x.dtype],
                n, c, h, w, ctypes.byref(_i64x4(x.stride())), s_w, s_h, sx, sy,
                float(gain), float(slope), float(min(clamp, 3.0e38)),
                1 if write_signs else (2 if read_signs else 0), _stream(x))
        _check(rc, 'filtered_lrelu_act_')
        return so


# ------------------------------------------------------------------------------------------------
# tri-plane gather / compositing / fused renderer / modulated conv / frame conversion
# ------------------------------------------------------------------------------------------------

class TriplanePlugin:
    @staticmethod
    def sample(planes, coords, ray_grid=None):
        """planes [n, 3C, H, W] float32 (any strides), coords [n, m, 3] float32 -> [n*m, C].

        ray_grid = (rays_h, rays_w, steps) tells the library that coords is [n, rays_h, rays_w, steps, 3] flattened (what
        the ray-marcher produces): a grouping hint for the LDS-staged kernel, results do not depend on it."""
        _require(planes.is_cuda and coords.device == planes.device, 'planes and coords must be on the same CUDA device')
        _require(planes.dtype == torch.float32 and coords.dtype == torch.float32, 'planes and coords must be float32')
        _require(planes.ndim == 4 and planes.shape[1] % 3 == 0, 'planes must be [n, 3*C, H, W]')
        _require(coords.ndim == 3 and coords.shape[2] == 3 and coords.shape[0] == planes.shape[0], 'coords must be [n, m, 3]')
        n, c3, H, W = planes.shape
        C = c3 // 3
        coords = coords.contiguous()
        m = coords.shape[1]
        out = torch.empty([n * m, C], dtype=torch.float32, device=planes.device)
        if m == 0:
            return out
        if ray_grid is not None:
            rh, rw, steps = (int(v) for v in ray_grid)
            _require(rh * rw * steps == m, 'ray_grid does not match the number of samples')
            with _dev_guard(planes.device):
                rc = load().ide3d_triplane_sample_rays(_ptr(planes), ctypes.byref(_i64x4(planes.stride())), n, C, H, W,
                                                       _ptr(coords), m, _ptr(out), rh, rw, steps, _stream(planes))
            _check(rc, 'triplane_sample_rays')
            return out
        with _dev_guard(planes.device):
            rc = load().ide3d_triplane_sample(_ptr(planes), ctypes.byref(_i64x4(planes.stride())), n, C, H, W,
                                              _ptr(coords), m, _ptr(out), _stream(planes))
        _check(rc, 'triplane_sample')
        return out

    @staticmethod
    def taps(H, W, coords):
        coords = coords.contiguous().reshape(-1, 3)
        _require(coords.is_cuda and coords.dtype == torch.float32, 'coords must be float32 on a CUDA device')
        taps = torch.empty([coords.shape[0], 3, 3], dtype=torch.int32, device=coords.device)
        if coords.shape[0]:
            with _dev_guard(coords.device):
                rc = load().ide3d_triplane_taps(H, W, _ptr(coords), coords.shape[0], _ptr(taps), _stream(coords))
            _check(rc, 'triplane_taps')
        return taps

    @staticmethod
    def sample_backward(grad_out, planes, coords, need_coord_grad):
        n, c3, H, W = planes.shape
        C = c3 // 3
        coords = coords.contiguous()
        grad_out = grad_out.contiguous()
        m = coords.shape[1]
        grad_planes = torch.zeros_like(planes)
        grad_coords = torch.zeros_like(coords) if need_coord_grad else None
        if m:
            with _dev_guard(planes.device):
                rc = load().ide3d_triplane_sample_backward(
                    _ptr(grad_out), _ptr(planes), ctypes.byref(_i64x4(planes.stride())), n, C, H, W,
                    _ptr(coords), m, _ptr(grad_planes), ctypes.byref(_i64x4(grad_planes.stride())),
                    _ptr(grad_coords), _stream(planes))
            _check(rc, 'triplane_sample_backward')
        return grad_planes, grad_coords


class VolumeRenderPlugin:
    @staticmethod
    def composite(rgb_sigma, z_vals, dir_norm, noise, clamp_mode, last_back, white_back, max_depth, fill_mode,
                  want_weights=True):
        """rgb_sigma [rays, steps, ch+1], z_vals [rays, steps], dir_norm [rays], noise [rays, steps] | None."""
        _require(rgb_sigma.is_cuda and rgb_sigma.dtype == torch.float32, 'rgb_sigma must be float32 on a CUDA device')
        rays, steps, row = rgb_sigma.shape
        ch = row - 1
        rgb_sigma, z_vals, dir_norm = rgb_sigma.contiguous(), z_vals.contiguous(), dir_norm.contiguous()
        if noise is not None:
            noise = noise.contiguous()
        dev = rgb_sigma.device
        rgb = torch.empty([rays, ch], dtype=torch.float32, device=dev)
        depth = torch.empty([rays], dtype=torch.float32, device=dev)
        weights = torch.empty([rays, steps], dtype=torch.float32, device=dev) if want_weights else None
        with _dev_guard(dev):
            rc = load().ide3d_composite(_ptr(rgb_sigma), _ptr(z_vals), _ptr(dir_norm), _ptr(noise), rays, steps, ch,
                                        int(clamp_mode), int(bool(last_back)), int(bool(white_back)), float(max_depth or 0.0),
                                        int(fill_mode), _ptr(rgb), _ptr(depth), _ptr(weights), _stream(rgb_sigma))
        _check(rc, 'composite')
        return rgb, depth, weights

    @staticmethod
    def sample_pdf(bins, weights, u, eps=1e-5):
        """bins [rays, k+1], weights [rays, k], u [n_importance] (shared) or [rays, n_importance] -> [rays, n_importance]."""
        _require(weights.is_cuda and weights.dtype == torch.float32 and weights.ndim == 2, 'weights must be 2-D float32 on a CUDA device')
        rays, k = weights.shape
        _require(bins.shape == (rays, k + 1) and bins.dtype == torch.float32 and bins.device == weights.device,
                 'bins must be [rays, k + 1] float32 next to weights')
        _require(u.dtype == torch.float32 and u.device == weights.device and u.ndim in (1, 2) and (u.ndim == 1 or u.shape[0] == rays),
                 'u must be [n_importance] or [rays, n_importance] float32 next to weights')
        bins, weights, u = bins.contiguous(), weights.contiguous(), u.contiguous()
        n_imp = u.shape[-1]
        out = torch.empty([rays, n_imp], dtype=torch.float32, device=weights.device)
        with _dev_guard(weights.device):
            rc = load().ide3d_sample_pdf(_ptr(bins), _ptr(weights), _ptr(u), 0 if u.ndim == 1 else n_imp, rays, k, n_imp,
                                         float(eps), _ptr(out), _stream(weights))
        _check(rc, 'sample_pdf')
        return out

    @staticmethod
    def _fill_render_params(p, tex_planes, geo_planes, mlp):
        p.tex_planes, p.geo_planes = tex_planes.data_ptr(), geo_planes.data_ptr()
        p.tex_stride, p.geo_stride = _i64x4(tex_planes.stride()), _i64x4(geo_planes.stride())
        for k in ('geo_w0', 'geo_b0', 'geo_w1', 'geo_b1', 'tex_w0', 'tex_b0', 'tex_w1', 'tex_b1'):
            t = mlp[k]
            _require(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), f'{k} must be contiguous float32 on GPU')
            setattr(p, k, t.data_ptr())
            _require(t.device == tex_planes.device, f'{k} must reside on the same device as the tri-planes')
        n, c3, H, W = tex_planes.shape
        _require(geo_planes.shape == tex_planes.shape, 'texture and geometry tri-planes must have the same shape')
        _require(geo_planes.device == tex_planes.device, 'texture and geometry tri-planes must reside on the same device')
        _require(c3 % 3 == 0, 'tri-planes must be [n, 3*C, H, W]')
        C = c3 // 3
        _require(mlp['geo_w0'].ndim == 2 and mlp['geo_w1'].ndim == 2 and mlp['tex_w0'].ndim == 2 and mlp['tex_w1'].ndim == 2,
                 'decoder weights must be [out, in] matrices')
        hidden = mlp['geo_w0'].shape[0]
        nout_geo, nout_tex = mlp['geo_w1'].shape[0], mlp['tex_w1'].shape[0]
        # the kernel indexes w0[row * C + col], b0[row], w1[row * hidden + col], b1[row] for the compiled (C, hidden): every
        # tensor must have exactly that shape, and both branches the same hidden width
        for k, shape in (('geo_w0', (hidden, C)), ('tex_w0', (hidden, C)), ('geo_b0', (hidden,)), ('tex_b0', (hidden,)),
                         ('geo_w1', (nout_geo, hidden)), ('tex_w1', (nout_tex, hidden)), ('geo_b1', (nout_geo,)), ('tex_b1', (nout_tex,))):
            _require(tuple(mlp[k].shape) == shape, f'{k} must be {list(shape)} (C={C}, hidden={hidden}), got {list(mlp[k].shape)}')
        _require(nout_geo >= 1, 'geo_w1 must have at least the sigma row')
        p.n, p.C, p.H, p.W = n, C, H, W
        p.hidden = hidden
        p.seg_ch = nout_geo - 1
        p.feat_ch = nout_tex

    @staticmethod
    def render_rays(rays_d_cam, z_lin, cam2world, jitter, sigma_noise, tex_planes, geo_planes, mlp,
                    clamp_mode, last_back, white_back, max_depth):
        dev = tex_planes.device
        for t in (rays_d_cam, z_lin, cam2world, tex_planes, geo_planes):
            _require(t.is_cuda and t.dtype == torch.float32, 'render_rays: float32 CUDA tensors required')
        rays_d_cam, z_lin = rays_d_cam.contiguous(), z_lin.contiguous()
        cam2world = cam2world.reshape(-1, 16).contiguous()
        p = _RenderParams()
        VolumeRenderPlugin._fill_render_params(p, tex_planes, geo_planes, mlp)
        p.rays_d_cam, p.z_lin, p.cam2world = rays_d_cam.data_ptr(), z_lin.data_ptr(), cam2world.data_ptr()
        p.rays_per_img, p.steps = rays_d_cam.shape[0], z_lin.shape[0]
        keep = [rays_d_cam, z_lin, cam2world]
        if jitter is not None:
            jitter = jitter.contiguous(); keep.append(jitter); p.jitter = jitter.data_ptr()
        if sigma_noise is not None:
            sigma_noise = sigma_noise.contiguous(); keep.append(sigma_noise); p.sigma_noise = sigma_noise.data_ptr()
        p.clamp_mode, p.last_back, p.white_back = int(clamp_mode), int(bool(last_back)), int(bool(white_back))
        p.max_depth = float(max_depth or 0.0)
        n, R = p.n, p.rays_per_img
        feat = torch.empty([n, p.feat_ch + p.seg_ch, R], dtype=torch.float32, device=dev)
        depth = torch.empty([n, R], dtype=torch.float32, device=dev)
        wsum = torch.empty([n, R], dtype=torch.float32, device=dev)
        p.out_feat, p.out_depth, p.out_wsum = feat.data_ptr(), depth.data_ptr(), wsum.data_ptr()
        with _dev_guard(dev):
            rc = load().ide3d_render_rays(ctypes.byref(p), _stream(tex_planes))
        if rc == -2:        # IDE3D_ENOKERNEL: configuration not covered by the fused kernel
            return None
        _check(rc, 'render_rays')
        return feat, depth, wsum

    @staticmethod
    def sample_voxel(tex_planes, geo_planes, mlp, pts, sigma_only=False):
        dev = tex_planes.device
        pts = pts.contiguous()
        _require(pts.is_cuda and pts.dtype == torch.float32 and pts.ndim == 3 and pts.shape[2] == 3, 'pts must be [n, m, 3] float32')
        p = _RenderParams()
        VolumeRenderPlugin._fill_render_params(p, tex_planes, geo_planes, mlp)
        n, m = pts.shape[0], pts.shape[1]
        _require(n == p.n, 'pts batch must match the tri-plane batch')
        width = p.feat_ch + p.seg_ch + 1
        out = None if sigma_only else torch.empty([n * m, width], dtype=torch.float32, device=dev)
        sig = torch.empty([n * m], dtype=torch.float32, device=dev) if sigma_only else None
        if m:
            with _dev_guard(dev):
                rc = load().ide3d_sample_voxel(ctypes.byref(p), _ptr(pts), m, _ptr(out), _ptr(sig), int(sigma_only), _stream(pts))
            if rc == -2:
                return None
            _check(rc, 'sample_voxel')
        return sig if sigma_only else out


    @staticmethod
    def _lattice(n, voxel_size, corner, scale):
        lat = _Lattice()
        lat.n, lat.voxel_size, lat.scale = int(n), float(voxel_size), float(scale)
        for i in range(3):
            lat.corner[i] = float(corner[i])
        return lat

    @staticmethod
    def lattice_points(n, voxel_size, corner, scale, first, count, device):
        """Points [first, first + count) of the extract_shapes lattice (see ide3d_lattice in the header) -> [count, 3]."""
        device = torch.device(device)
        _require(device.type == 'cuda', 'lattice_points: CUDA device required')
        pts = torch.empty([count, 3], dtype=torch.float32, device=device)
        lat = VolumeRenderPlugin._lattice(n, voxel_size, corner, scale)
        with _dev_guard(device):
            rc = load().ide3d_lattice_points(ctypes.byref(lat), int(first), int(count), _ptr(pts), _stream(pts))
        _check(rc, 'lattice_points')
        return pts

    @staticmethod
    def density_lattice(tex_planes, geo_planes, mlp, n, voxel_size, corner, scale, first, count):
        """sigma of lattice points [first, first + count) for every image -> [batch * count]; None if no fused kernel."""
        p = _RenderParams()
        VolumeRenderPlugin._fill_render_params(p, tex_planes, geo_planes, mlp)
        lat = VolumeRenderPlugin._lattice(n, voxel_size, corner, scale)
        sig = torch.empty([p.n * int(count)], dtype=torch.float32, device=tex_planes.device)
        if count:
            with _dev_guard(tex_planes.device):
                rc = load().ide3d_density_lattice(ctypes.byref(p), ctypes.byref(lat), int(first), int(count), _ptr(sig), _stream(tex_planes))
            if rc == -2:
                return None
            _check(rc, 'density_lattice')
        return sig


class ModconvPlugin:
    # workspace cache: (weight data_ptr, shape, n, h, w, mode) -> [buffer, weight._version the packed copy was made from, weakref(weight)]
    _ws = {}

    @staticmethod
    def modconv2d(x, w, styles, dcoefs, noise, noise_strength, bias, act, alpha, gain, clamp, mode=0, arith=0, x_amax=None, y_amax=None, pad_rows=False):
        """arith: 0 = process default (`conv_arithmetic`), 1 = fp32 MFMA, 3 = bf16x3, 6 = bf16x6, 16 = f16x3 (include/ide3d_hip.h).
        x_amax [n, AMAX_FLOATS]: row max = bound of max |x| per image (the f16x3 arithmetic needs it; else it runs bf16x6);
        y_amax [n, AMAX_FLOATS], zeroed: its row maxima receive max |finite y| per image.
        mode 0: stride-1 k x k (modulated) conv, "same" padding, fused epilogue; mode 1: 3x3 stride-2 conv without padding
        (output ((h-3)//2+1) x ((w-3)//2+1)); mode 2: 3x3 transposed stride-2 conv (output (2h+1) x (2w+1)).
        styles / dcoefs / noise / bias may be None."""
        for t in (x, w):
            _require(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), 'modconv2d: contiguous float32 CUDA tensors required')
        _require(w.device == x.device, 'modconv2d: w must reside on the same device as x')
        n, cin, h, wd = x.shape
        per_image = (w.ndim == 5)             # [n, cout, cin, k, k]: styles already folded into per-image weights
        if per_image:
            _require(w.shape[0] == n and styles is None, 'modconv2d: per-image weights need w.shape[0] == batch and styles=None')
        cout, cin2, k, k2 = w.shape[-4:]
        _require(cin == cin2 and k == k2 and k in (1, 3), 'modconv2d: weight must be [cout, cin, k, k] with k in {1, 3}')
        _require(mode in (0, 1, 2), 'modconv2d: mode must be 0, 1 or 2')
        _require(mode != 1 or (k == 3 and h >= 3 and wd >= 3), 'modconv2d: mode 1 is a 3x3 stride-2 convolution on an input of at least 3x3')
        oh, ow = (2 * h + 1, 2 * wd + 1) if mode == 2 else (((h - 3) // 2 + 1, (wd - 3) // 2 + 1) if mode == 1 else (h, wd))
        # pad_rows (mode 2): rows of the (2w + 1)-wide result padded to a multiple of 4 floats — returned as a view [..., :ow] of the padded
        # storage — so that the FIR that reads it next stages 16-byte aligned rows
        pitch = (ow + 3) // 4 * 4 if (pad_rows and mode == 2) else ow
        y = torch.empty([n, cout, oh, pitch], dtype=torch.float32, device=x.device)
        lib = load()
        # one workspace per (weight, problem shape, device, launch domain): the split-K partials inside it belong to one launch at a
        # time; the domain is the current stream for eager callers and the owning GraphedRenderer inside `workspace_scope`
        # ... and per arithmetic: the packed weights of the split-bf16 loops differ from the fp32 loop's
        arith = int(arith) or int(lib.ide3d_get_conv_arithmetic())
        if arith == 16 and x_amax is None:
            arith = 6          # f16x3 needs the bound on |x|; decided here so that the workspace (packed weights) is the bf16x6 one
        key = (0 if per_image else w.data_ptr(), tuple(w.shape), n, h, wd, mode, x.device.index, _ws_domain(x.device), arith)
        ent = ModconvPlugin._ws.get(key)
        if ent is None:
            nbytes = lib.ide3d_modconv_workspace_bytes(n, cin, cout, h, wd, k, mode, int(per_image))
            _require(nbytes >= 0, 'modconv2d: unsupported configuration')
            if len(ModconvPlugin._ws) > 1024:
                # drop only workspaces whose weight tensor is gone (nothing can launch with them again): a captured hipGraph
                # holds raw pointers into the live ones, so those are never freed behind its back
                for k_dead in [k_ for k_, e_ in ModconvPlugin._ws.items() if e_[2] is not None and e_[2]() is None]:
                    del ModconvPlugin._ws[k_dead]
            ent = [torch.empty([max(nbytes // 4, 1)], dtype=torch.float32, device=x.device), None, None]
            ModconvPlugin._ws[key] = ent
        p = _ModconvParams()
        p.x, p.w, p.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
        keep = []
        for name, t in (('styles', styles), ('dcoefs', dcoefs), ('noise', noise), ('bias', bias)):
            if t is not None:
                t = t.contiguous(); keep.append(t)
                _require(t.is_cuda and t.dtype == torch.float32, f'modconv2d: {name} must be float32 on GPU')
                _require(t.device == x.device, f'modconv2d: {name} must reside on the same device as x')
                setattr(p, name, t.data_ptr())
        p.n, p.cin, p.cout, p.h, p.w_, p.k = n, cin, cout, h, wd, k
        p.noise_strength = float(noise_strength)
        p.act, p.alpha, p.gain, p.clamp = int(act), float(alpha), float(gain), float(clamp)
        p.mode = mode
        # the packed copy in the workspace is valid only for the very tensor object (and version) it was made from: a data_ptr
        # can be recycled by the allocator for another weight of the same shape
        p.weights_packed = int((not per_image) and ent[2] is not None and ent[2]() is w and ent[1] == w._version)
        p.w_batch_stride = (cout * cin * k * k) if per_image else 0
        p.arith = arith
        for name, t in (('x_amax', x_amax), ('y_amax', y_amax)):
            if t is not None:
                _require(t.is_cuda and t.device == x.device and t.dtype == torch.float32 and tuple(t.shape) == (n, AMAX_FLOATS) and t.is_contiguous(),
                         f'modconv2d: {name} must be a contiguous float32 [n, AMAX_FLOATS] tensor on the device of x')
                setattr(p, name, t.data_ptr())
        p.workspace, p.workspace_bytes = ent[0].data_ptr(), ent[0].numel() * 4
        p.y_pitch = pitch if pitch != ow else 0
        with _dev_guard(x.device):
            rc = lib.ide3d_modconv2d(ctypes.byref(p), _stream(x))
        _check(rc, 'modconv2d')
        ent[1], ent[2] = (None, None) if per_image else (w._version, weakref.ref(w))
        return y if pitch == ow else y[..., :ow]


def modconv_plan(n, cin, cout, h, w, k=3, mode=0, per_image=False, arith=0, epilogue='conv', x_amax=False):
    """Host-only (works without a GPU): the kernel family, tile and grid `modconv2d` would launch for this shape -> dict of the
    ide3d_modconv_plan_info fields, `kind` as a name from PLAN_KINDS.  epilogue: 'conv' (noise, bias, lrelu, gain sqrt(2): the 3x3 layers),
    'plain' (demodulation only: the up-sampling layers, whose FIR carries the rest) or 'head' (bias, clamp 256, linear).  x_amax: the
    caller passes the producer's bound on |x| (what the f16x3 arithmetic needs)."""
    p = _ModconvParams()
    p.n, p.cin, p.cout, p.h, p.w_, p.k, p.mode = n, cin, cout, h, w, k, mode
    p.arith = int(arith)
    p.w_batch_stride = cout * cin * k * k if per_image else 0
    dummy = 256                                  # non-null, 16-byte aligned, never dereferenced
    p.x = p.w = p.y = dummy
    if not per_image:
        p.styles = p.dcoefs = dummy
    if epilogue == 'conv':
        p.noise, p.bias, p.noise_strength, p.act, p.alpha, p.gain, p.clamp = dummy, dummy, 1.0, 3, 0.2, math.sqrt(2.0), -1.0
    elif epilogue == 'plain':
        p.act, p.gain, p.clamp = 1, 1.0, -1.0
    elif epilogue == 'head':
        p.bias, p.act, p.gain, p.clamp = dummy, 1, 1.0, 256.0
    else:
        raise ValueError(epilogue)
    if x_amax:
        p.x_amax = dummy
    info = _ModconvPlanInfo()
    rc = load().ide3d_modconv_plan(ctypes.byref(p), ctypes.byref(info))       # (not a launch: stays out of CALLS)
    if rc != 0:
        raise RuntimeError(f'modconv_plan failed (code {rc}): ' + load().ide3d_last_error().decode('utf-8', 'replace'))
    out = {name: getattr(info, name) for name, _ in _ModconvPlanInfo._fields_ if name != 'reserved'}
    out['kind'] = PLAN_KINDS[info.kind]
    return out


_ARITH_NAMES = {'fp32': 1, 'bf16x3': 3, 'bf16x6': 6, 'f16x3': 16, 'default': 0}


def conv_arithmetic(name=None):
    """Get (no argument) or set the process-wide arithmetic of the shared-weight 3x3 convolutions (the per-image heads and the decoder MLPs of
    the ray-marcher follow it): 'bf16x6' (the default: every fp32 operand as 3 bf16 pieces, the 6 products above 2^-24 on the bf16 matrix
    pipe, fp32 accumulation - fp32-grade, measured as accurate as the fp32 matrix instruction against float64), 'fp32' (exact fp32
    products on v_mfma_f32_32x32x2_f32, 1.5x slower end to end), 'f16x3' (2 fp16 pieces, 3 products, power-of-two range scales from the
    producers' `amax`: ~2^-21 per product relative to the row / image maxima; the fastest fp32-grade mode), 'bf16x3' (2 bf16 pieces, 3
    products, ~2^-17 per product) or 'default' (back to the IDE3D_CONV_ARITH environment default, else bf16x6).  Returns the name in force.

    Safety beside other kernels: on MI355X a packed-fp32 instruction of ANOTHER kernel's wave returns wrong values
    while a wave on the same SIMD runs a loop of LDS reads + bf16 / fp16 MFMAs.  Every kernel of this library with such a loop keeps
    foreign waves off its SIMDs while the loop runs (8-wave workgroups that fill the register file of their SIMDs, or waves that claim
    all 512 registers), so every arithmetic may run beside kernels of other libraries, RCCL or another stream
    (tests/test_gpu_conv_arith.py::test_foreign_packed_fp32_victim_beside_every_matrix_loop)."""
    lib = load()
    if name is not None:
        _require(name in _ARITH_NAMES, f'conv_arithmetic: one of {sorted(_ARITH_NAMES)}')
        _check(lib.ide3d_set_conv_arithmetic(_ARITH_NAMES[name]), 'set_conv_arithmetic')
    code = int(lib.ide3d_get_conv_arithmetic())
    return {1: 'fp32', 3: 'bf16x3', 6: 'bf16x6', 16: 'f16x3'}[code]


class StylePlugin:
    @staticmethod
    def style_demod(w, affine_w, affine_b, affine_gain, bias_gain, wsq_t=None):
        """w [n, wdim] (rows may be strided), affine_w [cin, wdim], wsq_t [cin, cout] | None -> (styles [n, cin], dcoefs [n, cout] | None)."""
        _require(w.is_cuda and w.dtype == torch.float32 and w.ndim == 2 and w.stride(1) == 1, 'style_demod: w must be float32 [n, wdim] with unit inner stride')
        n, wdim = w.shape
        cin = affine_w.shape[0]
        _require(affine_w.is_contiguous() and affine_w.dtype == torch.float32 and affine_w.shape[1] == wdim, 'style_demod: bad affine weight')
        for name, t in (('affine_w', affine_w), ('affine_b', affine_b), ('wsq_t', wsq_t)):
            _require(t is None or t.device == w.device, f'style_demod: {name} must reside on the same device as w')
        styles = torch.empty([n, cin], dtype=torch.float32, device=w.device)
        dcoefs = None
        cout = 0
        if wsq_t is not None:
            _require(wsq_t.is_contiguous() and wsq_t.shape[0] == cin, 'style_demod: wsq_t must be contiguous [cin, cout]')
            cout = wsq_t.shape[1]
            dcoefs = torch.empty([n, cout], dtype=torch.float32, device=w.device)
        with _dev_guard(w.device):
            rc = load().ide3d_style_demod(_ptr(w), w.stride(0), _ptr(affine_w), _ptr(affine_b), _ptr(wsq_t), n, cin, cout, wdim,
                                          float(affine_gain), float(bias_gain), _ptr(styles), _ptr(dcoefs), _stream(w))
        _check(rc, 'style_demod')
        return styles, dcoefs

    @staticmethod
    def fold_heads(w, affine_gain, a0, b0, w0, gain0, a1, b1, w1, gain1):
        """-> per-image folded head weights [n, cout0 + cout1, cin, 1, 1]."""
        _require(w.is_cuda and w.dtype == torch.float32 and w.ndim == 2 and w.stride(1) == 1, 'fold_heads: w must be float32 [n, wdim]')
        n, wdim = w.shape
        cout0, cin = w0.shape[0], w0.shape[1]
        cout1 = w1.shape[0]
        for t in (a0, b0, w0, a1, b1, w1):
            _require(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), 'fold_heads: contiguous float32 CUDA tensors required')
            _require(t.device == w.device, 'fold_heads: all tensors must reside on the same device as w')
        out = torch.empty([n, cout0 + cout1, cin, 1, 1], dtype=torch.float32, device=w.device)
        with _dev_guard(w.device):
            rc = load().ide3d_fold_heads(_ptr(w), w.stride(0), n, cin, wdim, float(affine_gain), _ptr(a0), _ptr(b0), _ptr(w0), cout0, float(gain0),
                                         _ptr(a1), _ptr(b1), _ptr(w1), cout1, float(gain1), _ptr(out), _stream(w))
        _check(rc, 'fold_heads')
        return out

    @staticmethod
    def style_demod_batch(jobs):
        """All modulated layers of a pass in two launches.  jobs: [(w, affine_w, affine_b, affine_gain, bias_gain, wsq_t or None)]
        with w [n, wdim] (unit inner stride) -> [(styles [n, cin], dcoefs [n, cout] or None)], views of two buffers.  Same per-layer
        code as `style_demod`: bit-identical.  Returns None when the batch form does not apply (n > 8, > STYLE_BATCH_MAX layers handled
        by chunking)."""
        if not jobs:
            return []
        w0 = jobs[0][0]
        n, wdim = w0.shape
        if n > 8:
            return None
        dev = w0.device
        tot_s = sum(j[1].shape[0] for j in jobs)
        tot_d = sum(j[5].shape[1] for j in jobs if j[5] is not None)
        sbuf = torch.empty([n * tot_s], dtype=torch.float32, device=dev)
        dbuf = torch.empty([max(n * tot_d, 1)], dtype=torch.float32, device=dev)
        arr = (_StyleJob * len(jobs))()
        outs, so, do = [], 0, 0
        for k, (w, aw, ab, again, bgain, wsq_t) in enumerate(jobs):
            _require(w.is_cuda and w.dtype == torch.float32 and w.ndim == 2 and w.stride(1) == 1 and tuple(w.shape) == (n, wdim) and w.device == dev,
                     'style_demod_batch: every w must be float32 [n, wdim] with unit inner stride on one device')
            cin = aw.shape[0]
            _require(aw.is_contiguous() and aw.shape[1] == wdim and aw.dtype == torch.float32 and aw.device == dev, 'style_demod_batch: affine weight must be contiguous float32 [cin, wdim]')
            _require(ab is None or (ab.is_contiguous() and ab.numel() == cin and ab.device == dev), 'style_demod_batch: affine bias must be contiguous [cin]')
            styles = sbuf[so:so + n * cin].view(n, cin); so += n * cin
            dco, cout = None, 0
            if wsq_t is not None:
                _require(wsq_t.is_contiguous() and wsq_t.shape[0] == cin and wsq_t.device == dev, 'style_demod_batch: wsq_t must be contiguous [cin, cout]')
                cout = wsq_t.shape[1]
                dco = dbuf[do:do + n * cout].view(n, cout); do += n * cout
            j = arr[k]
            j.w, j.w_stride = w.data_ptr(), w.stride(0)
            j.affine_w, j.affine_b = aw.data_ptr(), (ab.data_ptr() if ab is not None else None)
            j.wsq_t = wsq_t.data_ptr() if wsq_t is not None else None
            j.cin, j.cout, j.affine_gain, j.bias_gain = cin, cout, float(again), float(bgain)
            j.styles, j.dcoefs = styles.data_ptr(), (dco.data_ptr() if dco is not None else None)
            outs.append((styles, dco))
        lib = load()
        with _dev_guard(dev):
            for k0 in range(0, len(jobs), STYLE_BATCH_MAX):
                cnt = min(STYLE_BATCH_MAX, len(jobs) - k0)
                rc = lib.ide3d_style_demod_batch(ctypes.cast(ctypes.byref(arr, k0 * ctypes.sizeof(_StyleJob)), ctypes.POINTER(_StyleJob)),
                                                 cnt, n, wdim, _stream(w0))
                _check(rc, 'style_demod_batch')
        return outs

    @staticmethod
    def fold_heads_batch(jobs):
        """All dual heads of a pass in one launch.  jobs: [(w, affine_gain, a0, b0, w0, gain0, a1, b1, w1, gain1)] (arguments of
        `fold_heads`) -> [out [n, cout0 + cout1, cin, 1, 1]], views of one buffer; bit-identical to `fold_heads`."""
        if not jobs:
            return []
        w_0 = jobs[0][0]
        n, wdim = w_0.shape
        dev = w_0.device
        sizes = [(j[4].shape[0] + j[8].shape[0]) * j[4].shape[1] for j in jobs]
        buf = torch.empty([n * sum(sizes)], dtype=torch.float32, device=dev)
        arr = (_FoldJob * len(jobs))()
        outs, off = [], 0
        for k, (w, again, a0, b0, w0, g0, a1, b1, w1, g1) in enumerate(jobs):
            _require(w.is_cuda and w.dtype == torch.float32 and w.ndim == 2 and w.stride(1) == 1 and tuple(w.shape) == (n, wdim) and w.device == dev,
                     'fold_heads_batch: every w must be float32 [n, wdim] with unit inner stride on one device')
            for t in (a0, b0, w0, a1, b1, w1):
                _require(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device == dev, 'fold_heads_batch: contiguous float32 CUDA tensors required')
            cout0, cin = w0.shape[0], w0.shape[1]
            cout1 = w1.shape[0]
            _require(w1.shape[1] == cin and a0.shape == (cin, wdim) and a1.shape == (cin, wdim), 'fold_heads_batch: head shapes do not agree')
            out = buf[off:off + n * sizes[k]].view(n, cout0 + cout1, cin, 1, 1); off += n * sizes[k]
            j = arr[k]
            j.w, j.w_stride, j.cin, j.affine_gain = w.data_ptr(), w.stride(0), cin, float(again)
            j.a0, j.b0, j.w0, j.cout0, j.gain0 = a0.data_ptr(), b0.data_ptr(), w0.data_ptr(), cout0, float(g0)
            j.a1, j.b1, j.w1, j.cout1, j.gain1 = a1.data_ptr(), b1.data_ptr(), w1.data_ptr(), cout1, float(g1)
            j.out = out.data_ptr()
            outs.append(out)
        lib = load()
        with _dev_guard(dev):
            for k0 in range(0, len(jobs), STYLE_BATCH_MAX):
                cnt = min(STYLE_BATCH_MAX, len(jobs) - k0)
                rc = lib.ide3d_fold_heads_batch(ctypes.cast(ctypes.byref(arr, k0 * ctypes.sizeof(_FoldJob)), ctypes.POINTER(_FoldJob)),
                                                cnt, n, wdim, _stream(w_0))
                _check(rc, 'fold_heads_batch')
        return outs


class FramePlugin:
    @staticmethod
    def frame_u8(img, seg, palette, out=None):
        img, seg = img.contiguous(), seg.contiguous()
        _require(img.is_cuda and img.dtype == torch.float32 and seg.dtype == torch.float32, 'frame_u8: float32 CUDA tensors required')
        n, _, H, W = img.shape
        classes = seg.shape[1]
        _require(palette.dtype == torch.uint8 and palette.shape == (classes, 3) and palette.is_cuda, 'palette must be uint8 [classes, 3] on GPU')
        if out is None:
            out = torch.empty([n, H, 2 * W, 3], dtype=torch.uint8, device=img.device)
        _require(out.dtype == torch.uint8 and tuple(out.shape) == (n, H, 2 * W, 3) and out.is_contiguous() and out.device == img.device,
                 'frame_u8: out must be a contiguous uint8 [N, H, 2W, 3] tensor on the image device')
        with _dev_guard(img.device):
            rc = load().ide3d_frame_u8(_ptr(img), _ptr(seg), _ptr(palette.contiguous()), n, classes, H, W, _ptr(out), _stream(img))
        _check(rc, 'frame_u8')
        return out


class CameraPlugin:
    """The pose helpers of training/volumetric_rendering.py as two launches (csrc/camera.hip)."""

    @staticmethod
    def applies(*tensors):
        """float32 CUDA tensors outside autograd (the drivers' poses; a differentiable pose keeps the tensor operations)."""
        return all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and not t.requires_grad for t in tensors)

    @staticmethod
    def sphere_points(theta, pitch, r, pitch_is_v=False):
        """theta, pitch [n, 1] -> (pos [n, 3], phi [n, 1]) (include/ide3d_hip.h: ide3d_sphere_points)."""
        _require(theta.is_cuda and theta.dtype == torch.float32 and pitch.dtype == torch.float32 and pitch.device == theta.device and theta.numel() == pitch.numel(),
                 'sphere_points: float32 CUDA tensors of one size required')
        theta, pitch = theta.contiguous(), pitch.contiguous()
        n = theta.numel()
        pos = torch.empty([n, 3], dtype=torch.float32, device=theta.device)
        phi = torch.empty_like(pitch)
        with _dev_guard(theta.device):
            rc = load().ide3d_sphere_points(_ptr(theta), _ptr(pitch), n, float(r), int(bool(pitch_is_v)), _ptr(pos), _ptr(phi), _stream(theta))
        _check(rc, 'sphere_points')
        return pos, phi

    @staticmethod
    def cam2world(forward, origin, lookat=None):
        """forward [n, 3] | None, origin [n, 3], lookat [3] / [1, 3] / [n, 3] | None -> [n, 4, 4] (ide3d_cam2world)."""
        _require(origin.is_cuda and origin.dtype == torch.float32 and origin.ndim == 2 and origin.shape[1] == 3, 'cam2world: origin must be float32 [n, 3] on a CUDA device')
        origin = origin.contiguous()
        n = origin.shape[0]
        stride = 0
        if lookat is not None:
            _require(lookat.dtype == torch.float32 and lookat.device == origin.device and lookat.numel() in (3, 3 * n), 'cam2world: lookat must hold 3 or 3 n float32 values on the same device')
            lookat = lookat.contiguous()
            stride = 3 if (lookat.numel() == 3 * n and n > 1) else 0
        else:
            _require(forward is not None and forward.dtype == torch.float32 and forward.device == origin.device and tuple(forward.shape) == (n, 3),
                     'cam2world: forward must be float32 [n, 3] on the same device')
            forward = forward.contiguous()
        out = torch.empty([n, 4, 4], dtype=torch.float32, device=origin.device)
        with _dev_guard(origin.device):
            rc = load().ide3d_cam2world(_ptr(forward if lookat is None else None), _ptr(origin), _ptr(lookat), stride, n, _ptr(out), _stream(origin))
        _check(rc, 'cam2world')
        return out


class MappingPlugin:
    MAX_N, MAX_WIDTH, MAX_LAYERS = 8, 1024, 16
    _ws = {}          # (device index, launch domain) -> workspace tensor (per-layer activations + barrier counter); see workspace_scope

    @staticmethod
    def supports(n, z_dim, embed, widths, device=None):
        """`device`: the device the launch will run on (the residency answer is per device; None = the current one)."""
        k0 = z_dim + embed
        if not (1 <= n <= MappingPlugin.MAX_N and 0 < k0 <= MappingPlugin.MAX_WIDTH and k0 % 4 == 0 and 1 <= len(widths) <= MappingPlugin.MAX_LAYERS
                and all(0 < w <= MappingPlugin.MAX_WIDTH and w % 4 == 0 for w in widths)):
            return False
        with _dev_guard(device):                              # None: no-op guard
            return bool(load().ide3d_mapping_supported())            # the kernel's grid barrier needs its 64 workgroups co-resident

    @staticmethod
    def mapping(z, c, embed_w, embed_b, embed_wgain, embed_bgain, fc_ws, fc_bs, lr_multiplier, alpha, act_gain, num_ws, w_avg, psi, cutoff, trusted=False):
        """z [n, z_dim], c [n, c_dim] | None, raw parameters of the embed / fc layers -> ws [n, num_ws, w_dim] (one launch).
        `trusted`: the caller has passed these very parameter objects (and this batch size) through the checks before; only z / c are checked."""
        dev = z.device
        if trusted:
            _require(z.is_cuda and z.dtype == torch.float32 and z.is_contiguous() and (c is None or (c.device == dev and c.dtype == torch.float32 and c.is_contiguous()
                                                                                                 and tuple(c.shape) == (z.shape[0], embed_w.shape[1]))),
                     'mapping: z / c must be contiguous float32 tensors on one CUDA device, c [n, c_dim]')
            return MappingPlugin._launch(z, c, embed_w, embed_b, embed_wgain, embed_bgain, fc_ws, fc_bs, lr_multiplier, alpha, act_gain, num_ws, w_avg, psi, cutoff)
        f32c = lambda t: t is None or (t.is_cuda and t.device == dev and t.dtype == torch.float32 and t.is_contiguous())
        _require(all(f32c(t) for t in (z, c, embed_w, embed_b, w_avg, *fc_ws, *fc_bs)), 'mapping: contiguous float32 tensors on one CUDA device required')
        n, z_dim = z.shape
        embed = 0 if embed_w is None else embed_w.shape[0]
        _require(MappingPlugin.supports(n, z_dim, embed, [w.shape[0] for w in fc_ws], device=dev), 'mapping: unsupported shape')
        # the kernel indexes c as [n, c_dim] and w_avg as [w_dim]: a smaller conditioning batch (torch.cat would raise in the
        # reference, networks.py:302) or a short w_avg must not become an out-of-bounds read
        _require((embed_w is None) or (c is not None and embed_w.ndim == 2 and tuple(c.shape) == (n, embed_w.shape[1])),
                 f'mapping: c must be [{n}, {None if embed_w is None else embed_w.shape[1]}] (got {None if c is None else tuple(c.shape)})')
        _require(embed_b is None or (embed_w is not None and tuple(embed_b.shape) == (embed,)), 'mapping: embed bias must be [embed]')
        _require(w_avg is None or w_avg.numel() == fc_ws[-1].shape[0], 'mapping: w_avg must have w_dim elements')
        _require(num_ws >= 1, 'mapping: num_ws must be positive')
        k = z_dim + embed
        for w, b in zip(fc_ws, fc_bs):
            _require(w.ndim == 2 and w.shape[1] == k and (b is None or tuple(b.shape) == (w.shape[0],)), 'mapping: layer widths do not chain')
            k = w.shape[0]
        return MappingPlugin._launch(z, c, embed_w, embed_b, embed_wgain, embed_bgain, fc_ws, fc_bs, lr_multiplier, alpha, act_gain, num_ws, w_avg, psi, cutoff)

    @staticmethod
    def _launch(z, c, embed_w, embed_b, embed_wgain, embed_bgain, fc_ws, fc_bs, lr_multiplier, alpha, act_gain, num_ws, w_avg, psi, cutoff):
        dev = z.device
        n, z_dim = z.shape
        embed = 0 if embed_w is None else embed_w.shape[0]
        k = fc_ws[-1].shape[0]
        lib = load()
        key = (dev.index, _ws_domain(dev))
        wsp = MappingPlugin._ws.get(key)
        if wsp is None:
            wsp = MappingPlugin._ws[key] = torch.zeros([lib.ide3d_mapping_workspace_bytes() // 4 + 4], dtype=torch.float32, device=dev)
        out = torch.empty([n, num_ws, k], dtype=torch.float32, device=dev)
        p = _MappingParams()
        p.z, p.c = z.data_ptr(), (c.data_ptr() if c is not None else 0)
        p.embed_w, p.embed_b = (embed_w.data_ptr() if embed_w is not None else 0), (embed_b.data_ptr() if embed_b is not None else 0)
        for i, (w, b) in enumerate(zip(fc_ws, fc_bs)):
            p.fc_w[i], p.fc_b[i], p.fc_out[i] = w.data_ptr(), (b.data_ptr() if b is not None else 0), w.shape[0]
        p.w_avg = w_avg.data_ptr() if w_avg is not None else 0
        p.ws, p.workspace, p.workspace_bytes = out.data_ptr(), wsp.data_ptr(), wsp.numel() * 4
        p.n, p.z_dim, p.c_dim, p.embed, p.layers, p.num_ws = n, z_dim, (c.shape[1] if c is not None else 0), embed, len(fc_ws), num_ws
        p.embed_weight_gain, p.embed_bias_gain, p.lr_multiplier = float(embed_wgain), float(embed_bgain), float(lr_multiplier)
        p.alpha, p.act_gain, p.truncation_psi = float(alpha), float(act_gain), float(psi)
        p.truncation_cutoff = -1 if cutoff is None else int(cutoff)
        with _dev_guard(dev):
            rc = lib.ide3d_mapping(ctypes.byref(p), _stream(z))
        _check(rc, 'mapping')
        return out


class ResamplePlugin:
    @staticmethod
    def skip_upsample_add_cl(lo, add):
        """upsample2d(lo, [1,3,3,1]) + add -> [n, c, 2h, 2w] in channels_last memory format (lo / add: any strides)."""
        _require(lo.is_cuda and lo.dtype == torch.float32 and add.dtype == torch.float32 and add.device == lo.device,
                 'skip_upsample_add_cl: float32 CUDA tensors on one device required')
        n, c, h, w = lo.shape
        _require(tuple(add.shape) == (n, c, 2 * h, 2 * w), 'skip_upsample_add_cl: add must be [n, c, 2h, 2w]')
        _require(c % 4 == 0, 'skip_upsample_add_cl: channel count must be a multiple of 4')
        out = torch.empty([n, c, 2 * h, 2 * w], dtype=torch.float32, device=lo.device, memory_format=torch.channels_last)
        with _dev_guard(lo.device):
            rc = load().ide3d_skip_upsample_add_cl(_ptr(lo), ctypes.byref(_i64x4(lo.stride())), _ptr(add), ctypes.byref(_i64x4(add.stride())),
                                                   n, c, h, w, _ptr(out), _stream(lo))
        _check(rc, 'skip_upsample_add_cl')
        return out

    @staticmethod
    def bilinear_up2_split(x, ranges, adjacent=None):
        """x [n, c, h, w] -> one [n, count, 2h, 2w] tensor per (begin, count) in `ranges` (at most 3), bilinear, align_corners=False.
        `adjacent=(i, j)`: outputs i and j = i + 1 are the two channel ranges of ONE [n, count_i + count_j, 2h, 2w] tensor (views)."""
        _require(x.is_cuda and x.dtype == torch.float32, 'bilinear_up2_split: float32 CUDA tensor required')
        _require(1 <= len(ranges) <= 3, 'bilinear_up2_split: one to three channel ranges')
        x = x.contiguous()
        n, c, h, w = x.shape
        outs = [None if (adjacent and k in adjacent) else torch.empty([n, cnt, 2 * h, 2 * w], dtype=torch.float32, device=x.device) for k, (_b, cnt) in enumerate(ranges)]
        bs = [0, 0, 0]
        if adjacent:
            i, j = adjacent
            _require(j == i + 1 and 0 <= i and j < len(ranges), 'bilinear_up2_split: adjacent outputs must be consecutive')
            ci, cj = ranges[i][1], ranges[j][1]
            both = torch.empty([n, ci + cj, 2 * h, 2 * w], dtype=torch.float32, device=x.device)
            outs[i], outs[j] = both[:, :ci], both[:, ci:]
            bs[i] = bs[j] = (ci + cj) * 4 * h * w
        dst = (ctypes.c_void_p * 3)(*([o.data_ptr() for o in outs] + [0] * (3 - len(outs))))
        bsa = (ctypes.c_int64 * 3)(*bs)
        beg = (ctypes.c_int32 * 3)(*([int(b) for b, _c in ranges] + [0] * (3 - len(outs))))
        cnt = (ctypes.c_int32 * 3)(*([int(c_) for _b, c_ in ranges] + [0] * (3 - len(outs))))
        with _dev_guard(x.device):
            rc = load().ide3d_bilinear_up2_split(_ptr(x), n, c, h, w, ctypes.byref(dst), ctypes.byref(beg), ctypes.byref(cnt), ctypes.byref(bsa), _stream(x))
        _check(rc, 'bilinear_up2_split')
        return outs


class LowresPlugin:
    """The low-resolution block group of the backbone in one launch (csrc/lowres.hip, include/ide3d_hip.h `ide3d_lowres_group`)."""
    # workspace cache: key -> [buffer (zero-filled once: the halo slots of the activation images are never written), per layer (weakref(weight), version)]
    _ws = {}

    @staticmethod
    def layers_supported(n, C, res0, ups, arith=0):
        arr = (ctypes.c_int32 * len(ups))(*[int(u) for u in ups])
        return int(load().ide3d_lowres_layers_supported(int(n), int(C), int(res0), arr, len(ups), int(arith)))

    @staticmethod
    def persistent_default():
        return os.environ.get('IDE3D_LOWRES_PERSISTENT', '0') == '1'

    @staticmethod
    def group(x0, layers, heads, fir, persistent=None, arith=0):
        """x0 [C, r, r] (shared by the batch) or [n, C, r, r]; layers: dicts(weight [C, C, 3, 3], styles [n, C], dcoefs [n, C], noise [res, res] | None
        (x noise_strength already), bias [C] | None, act_gain, clamp (< 0: none), up (1 | 2), head (index | -1)); heads: dicts(w [n, O, C], bias [O] |
        None, clamp).  Returns (x_out [n, C, res, res], [skip_k [n, O, res_k, res_k]])."""
        lib = load()
        n, C = layers[0]['styles'].shape
        dev = x0.device
        _require(1 <= len(layers) <= LOWRES_MAX_LAYERS and len(heads) <= LOWRES_MAX_HEADS, 'lowres_group: too many layers / heads')
        _require(x0.is_cuda and x0.dtype == torch.float32 and x0.is_contiguous() and x0.shape[-3] == C and x0.shape[-1] == x0.shape[-2], 'lowres_group: x0 must be contiguous float32 [C, r, r] or [n, C, r, r]')
        _require(fir.is_cuda and fir.dtype == torch.float32 and fir.is_contiguous() and tuple(fir.shape) == (4, 4) and fir.device == dev, 'lowres_group: fir must be a float32 [4, 4] filter on the device of x0')
        arith = int(arith) or int(lib.ide3d_get_conv_arithmetic())
        p = _LowresParams()
        p.x0, p.x0_batch_stride = x0.data_ptr(), (x0.stride(0) if x0.ndim == 4 else 0)
        p.fir = fir.data_ptr()
        p.n, p.C, p.res0, p.nlayers, p.nheads, p.arith = n, C, x0.shape[-1], len(layers), len(heads), arith
        p.persistent = int(LowresPlugin.persistent_default() if persistent is None else bool(persistent))
        keep, res, head_res = [], x0.shape[-1], {}
        for l, L in enumerate(layers):
            q = p.layers[l]
            w = L['weight']
            _require(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape) == (C, C, 3, 3) and w.device == dev, f'lowres_group: layer {l}: weight must be contiguous float32 [C, C, 3, 3]')
            q.weight = w.data_ptr()
            res = res * 2 if int(L['up']) == 2 else res
            for name, shape in (('styles', (n, C)), ('dcoefs', (n, C)), ('noise', (res, res)), ('bias', (C,))):
                t = L.get(name)
                if t is None:
                    _require(name in ('noise', 'bias'), f'lowres_group: layer {l}: {name} is required')
                    continue
                t = t.contiguous(); keep.append(t)
                _require(t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == shape and t.device == dev, f'lowres_group: layer {l}: {name} must be float32 {shape} on the device of x0')
                setattr(q, name, t.data_ptr())
            q.act_gain, q.clamp, q.up, q.head = float(L['act_gain']), float(L['clamp']), int(L['up']), int(L.get('head', -1))
            if q.head >= 0:
                head_res[q.head] = res
        key = (tuple(L['weight'].data_ptr() for L in layers), n, C, x0.shape[-1], tuple(int(L['up']) for L in layers), dev.index, _ws_domain(dev), arith)
        ent = LowresPlugin._ws.get(key)
        if ent is None:
            nbytes = lib.ide3d_lowres_workspace_bytes(ctypes.byref(p))
            _require(nbytes > 0, 'lowres_group: unsupported configuration')
            if len(LowresPlugin._ws) > 64:
                for k_dead in [k_ for k_, e_ in LowresPlugin._ws.items() if any(r() is None for r, _ in e_[1])]:
                    del LowresPlugin._ws[k_dead]
            ent = LowresPlugin._ws[key] = [torch.zeros([nbytes // 4 + 1], dtype=torch.float32, device=dev), [(lambda: None, None)] * len(layers)]
        for l, L in enumerate(layers):
            ref, ver = ent[1][l]
            p.layers[l].weights_packed = int(ref() is L['weight'] and ver == L['weight']._version)
        p.workspace, p.workspace_bytes = ent[0].data_ptr(), ent[0].numel() * 4
        skips = []
        for k, H in enumerate(heads):
            q = p.heads[k]
            w = H['w'].reshape(n, -1, C)
            _require(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.device == dev, f'lowres_group: head {k}: folded weights must be contiguous float32 [n, O, C]')
            _require(k in head_res, f'lowres_group: head {k} is not attached to a layer')
            O = w.shape[1]
            keep.append(w)
            q.w, q.O, q.clamp = w.data_ptr(), O, float(H['clamp'])
            b = H.get('bias')
            if b is not None:
                b = b.contiguous(); keep.append(b)
                _require(b.is_cuda and b.dtype == torch.float32 and tuple(b.shape) == (O,) and b.device == dev, f'lowres_group: head {k}: bias must be float32 [O]')
                q.bias = b.data_ptr()
            sk = torch.empty([n, O, head_res[k], head_res[k]], dtype=torch.float32, device=dev)
            q.skip = sk.data_ptr()
            skips.append(sk)
        x_out = torch.empty([n, C, res, res], dtype=torch.float32, device=dev)
        p.x_out = x_out.data_ptr()
        with _dev_guard(dev):
            rc = lib.ide3d_lowres_group(ctypes.byref(p), _stream(x0))
        _check(rc, 'lowres_group')
        ent[1] = [(weakref.ref(L['weight']), L['weight']._version) for L in layers]
        return x_out, skips


PLUGINS = {
    'lowres_plugin': LowresPlugin,
    'bias_act_plugin': BiasActPlugin,
    'upfirdn2d_plugin': Upfirdn2dPlugin,
    'filtered_lrelu_plugin': FilteredLReluPlugin,
    'triplane_plugin': TriplanePlugin,
    'volume_render_plugin': VolumeRenderPlugin,
    'modconv_plugin': ModconvPlugin,
    'frame_plugin': FramePlugin,
    'camera_plugin': CameraPlugin,
    'style_plugin': StylePlugin,
    'resample_plugin': ResamplePlugin,
    'mapping_plugin': MappingPlugin,
}
