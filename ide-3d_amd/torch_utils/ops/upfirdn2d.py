"""2-D resampling with FIR filters (`torch_utils.ops.upfirdn2d` surface of the reference, upfirdn2d.py).

Public names, argument meaning and defaults follow the reference: `setup_filter` (:70), `upfirdn2d`
(:118), `filter2d` (:277), `upsample2d` (:313), `downsample2d` (:352), plus `_parse_padding` /
`_get_filter_size`, which `conv2d_resample` imports (conv2d_resample.py:16-17).  Device tensors
run `csrc/upfirdn2d.hip` through the C ABI; CPU tensors or `impl='ref'` run the PyTorch definition.
"""

import collections

import numpy as np
import torch

from .. import custom_ops
from .. import misc
from . import conv2d_gradfix

_plugin = None


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(module_name='upfirdn2d_plugin', sources=['upfirdn2d.hip'], headers=['common.h'])
    return True


def _int_pair(v, what):
    vals = [v, v] if isinstance(v, int) else list(v)
    assert len(vals) == 2 and all(isinstance(e, int) for e in vals), f'{what} must be an int or a pair of ints'
    return vals


def _parse_scaling(scaling):
    """int | [sx, sy] -> (sx, sy), both >= 1."""
    sx, sy = _int_pair(scaling, 'scaling')
    assert min(sx, sy) >= 1
    return sx, sy


def _parse_padding(padding):
    """int | [px, py] | [px0, px1, py0, py1] -> (px0, px1, py0, py1); negative values crop."""
    vals = [padding] * 2 if isinstance(padding, int) else list(padding)
    assert all(isinstance(e, int) for e in vals) and len(vals) in (2, 4), 'padding must be an int, a pair or four ints'
    if len(vals) == 2:
        vals = [vals[0], vals[0], vals[1], vals[1]]
    return tuple(vals)


def _get_filter_size(f):
    """(width, height) of a filter; None counts as 1 x 1, a 1-D filter is separable (same taps on both axes)."""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in (1, 2)
    with misc.suppress_tracer_warnings():
        fw, fh = int(f.shape[-1]), int(f.shape[0])
    misc.assert_shape(f, [fh, fw][:f.ndim])
    assert min(fw, fh) >= 1
    return fw, fh


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Prepare a FIR filter for `upfirdn2d()`; float32 [h, w] (non-separable) or [taps] (separable).

    Same rules as the reference (upfirdn2d.py:91-114): scalars become 1-tap filters, 1-D filters with
    fewer than 8 taps are expanded to their outer product, the filter is normalised to unit DC gain,
    optionally flipped, and scaled by `gain ** (ndim / 2)`.
    """
    f = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2]
    assert f.numel() > 0
    if f.ndim == 0:
        f = f.reshape(1)
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Pad, zero-upsample, FIR-filter and decimate a batch of images [N, C, H, W] (reference :118-162)."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _Upfirdn2dHip.apply(x, f, _Resample(*_parse_scaling(up), *_parse_scaling(down), *_parse_padding(padding), bool(flip_filter), gain))
    return _upfirdn2d_ref(x, f, up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain)


@misc.profiled_function
def _upfirdn2d_ref(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """PyTorch definition of the op (CPU path; reference upfirdn2d.py:167)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    assert f.dtype == torch.float32 and not f.requires_grad
    n, c, ih, iw = x.shape
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    assert iw * upx + padx0 + padx1 >= f.shape[-1] and ih * upy + pady0 + pady1 >= f.shape[0]

    # zero-stuffing upsample
    u = x.new_zeros([n, c, ih, upy, iw, upx])
    u[:, :, :, 0, :, 0] = x
    u = u.reshape(n, c, ih * upy, iw * upx)
    # positive pads add zeros, negative pads crop
    u = torch.nn.functional.pad(u, [max(padx0, 0), max(padx1, 0), max(pady0, 0), max(pady1, 0)])
    u = u[:, :, max(-pady0, 0): u.shape[2] - max(-pady1, 0), max(-padx0, 0): u.shape[3] - max(-padx1, 0)]

    k = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:
        k = k.flip(list(range(k.ndim)))
    if k.ndim == 2:
        u = conv2d_gradfix.conv2d(input=u, weight=k.expand(c, 1, *k.shape), groups=c)
    else:
        u = conv2d_gradfix.conv2d(input=u, weight=k.reshape(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)
        u = conv2d_gradfix.conv2d(input=u, weight=k.reshape(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)
    return u[:, :, ::downy, ::downx]


# resolved parameters of one device call (hashable; handed to the autograd function as a plain argument)
_Resample = collections.namedtuple('_Resample', 'upx upy downx downy px0 px1 py0 py1 flip gain')


class _Upfirdn2dHip(torch.autograd.Function):
    """`ide3d_upfirdn2d` forward; the adjoint of "upsample, pad, filter, decimate" is the same pipeline with the two
    factors exchanged, the filter mirrored and the padding chosen so that the input size comes back (reference :250-269)."""

    @staticmethod
    def forward(ctx, x, f, r):
        assert isinstance(x, torch.Tensor) and x.ndim == 4
        if f is None:
            f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
        elif f.ndim == 1 and f.shape[0] == 1:
            f = f.square().unsqueeze(0)           # a 1-tap separable filter is the 1x1 filter with the squared tap
        assert isinstance(f, torch.Tensor) and f.ndim in (1, 2)
        if f.ndim == 2:
            y = _plugin.upfirdn2d(x, f, r.upx, r.upy, r.downx, r.downy, r.px0, r.px1, r.py0, r.py1, r.flip, r.gain)
        else:
            # separable taps: rows first, then columns (the gain rides on the second pass)
            y = _plugin.upfirdn2d(x, f.unsqueeze(0), r.upx, 1, r.downx, 1, r.px0, r.px1, 0, 0, r.flip, 1.0)
            y = _plugin.upfirdn2d(y, f.unsqueeze(1), 1, r.upy, 1, r.downy, 0, 0, r.py0, r.py1, r.flip, r.gain)
        ctx.save_for_backward(f)
        ctx.r, ctx.in_hw = r, (x.shape[2], x.shape[3])
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError('upfirdn2d: the filter is not differentiable')
        if not ctx.needs_input_grad[0]:
            return None, None, None
        (f,) = ctx.saved_tensors
        r, (ih, iw) = ctx.r, ctx.in_hw
        fw, fh = _get_filter_size(f)
        back = _Resample(upx=r.downx, upy=r.downy, downx=r.upx, downy=r.upy,
                         px0=fw - r.px0 - 1, px1=iw * r.upx - dy.shape[3] * r.downx + r.px0 - r.upx + 1,
                         py0=fh - r.py0 - 1, py1=ih * r.upy - dy.shape[2] * r.downy + r.py0 - r.upy + 1,
                         flip=not r.flip, gain=r.gain)
        return _Upfirdn2dHip.apply(dy, f, back), None, None


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Filter with `f`, output the same size as the input (+ user padding) (reference :277-309)."""
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + fw // 2, padx1 + (fw - 1) // 2, pady0 + fh // 2, pady1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Upsample by `up`, output size a multiple of the input; gain is scaled by up^2 (reference :313-348)."""
    upx, upy = _parse_scaling(up)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw + upx - 1) // 2, padx1 + (fw - upx) // 2, pady0 + (fh + upy - 1) // 2, pady1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Downsample by `down`, output size a fraction of the input (reference :352-387)."""
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw - downx + 1) // 2, padx1 + (fw - downx) // 2, pady0 + (fh - downy + 1) // 2, pady1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
