"""Filtered leaky ReLU — the `torch_utils.ops.filtered_lrelu.filtered_lrelu` call surface (reference filtered_lrelu.py:56).

    y = down_fir( clamp( gain * lrelu_slope( up_fir( x + b ) * up^2 ) ) )

One HIP launch (`csrc/filtered_lrelu.hip`) for device tensors; the backward pass is the same launch with the two filters
swapped, the padding mirrored and the activation replaced by the 2-bit sign / clamp codes the forward pass stored
(reference :238-268).  Configurations the fused kernel declines (`IDE3D_ENOKERNEL`, the reference's `return_code = -1`,
:223-229) run as upfirdn2d -> in-place sign-coded activation -> upfirdn2d, still on HIP kernels.  CPU tensors and
`impl='ref'` evaluate the definition with the separate ops, as the reference does (:121-154).
"""

import collections
import warnings

import numpy as np
import torch

from .. import custom_ops
from .. import misc
from . import bias_act
from . import upfirdn2d

_plugin = None

# static configuration of one call (hashable: autograd functions receive it as a plain argument)
_Config = collections.namedtuple('_Config', 'up down px0 px1 py0 py1 gain slope clamp flip')


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(module_name='filtered_lrelu_plugin', sources=['filtered_lrelu.hip'], headers=['common.h'])
    return True


def _filter_wh(f):
    """(width, height) of a FIR given as None / 1-D separable / 2-D tensor."""
    if f is None:
        return 1, 1
    if not (isinstance(f, torch.Tensor) and f.ndim in (1, 2)):
        raise AssertionError('filter must be a 1-D or 2-D tensor')
    return int(f.shape[-1]), int(f.shape[0])


def _pad4(padding):
    """int | [px, py] | [px0, px1, py0, py1] -> four ints."""
    vals = [padding] * 2 if isinstance(padding, (int, np.integer)) else list(padding)
    if len(vals) == 2:
        vals = [vals[0], vals[0], vals[1], vals[1]]
    if len(vals) != 4 or not all(isinstance(v, (int, np.integer)) for v in vals):
        raise AssertionError('padding must be an int, [x, y] or [x0, x1, y0, y1] of ints')
    return tuple(int(v) for v in vals)


def _config(up, down, padding, gain, slope, clamp, flip_filter):
    ok = isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    ok = ok and float(gain) == gain and gain > 0 and float(slope) == slope and slope >= 0
    ok = ok and (clamp is None or (float(clamp) == clamp and clamp >= 0))
    if not ok:
        raise AssertionError('filtered_lrelu: up / down must be positive ints, gain > 0, slope >= 0, clamp None or >= 0')
    return _Config(up, down, *_pad4(padding), float(gain), float(slope), float('inf') if clamp is None else float(clamp), bool(flip_filter))


def _output_hw(x, fu, fd, cfg):
    (uw, uh), (dw, dh) = _filter_wh(fu), _filter_wh(fd)
    ow = (x.shape[3] * cfg.up + cfg.px0 + cfg.px1 - (uw - 1) - (dw - 1) + cfg.down - 1) // cfg.down
    oh = (x.shape[2] * cfg.up + cfg.py0 + cfg.py1 - (uh - 1) - (dh - 1) + cfg.down - 1) // cfg.down
    return oh, ow


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, impl='cuda'):
    """x [N, C, H, W]; fu / fd up- / down-sampling FIRs (None = identity, 1-D = separable); b [C] or None.
    Arguments, defaults and result as in the reference (filtered_lrelu.py:56-116)."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        cfg = _config(up, down, padding, gain, slope, clamp, flip_filter)
        return _FilteredLReluHip.apply(x, fu, fd, b, None, 0, 0, cfg)
    return _filtered_lrelu_ref(x, fu=fu, fd=fd, b=b, up=up, down=down, padding=padding, gain=gain, slope=slope, clamp=clamp,
                               flip_filter=flip_filter)


@misc.profiled_function
def _filtered_lrelu_ref(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None,
                        flip_filter=False):
    """The definition, spelled with bias_act / upfirdn2d (what CPU tensors run; reference :121-154)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    cfg = _config(up, down, padding, gain, slope, clamp, flip_filter)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.dtype == x.dtype
        misc.assert_shape(b, [x.shape[1]])
    want = [x.shape[0], x.shape[1], *_output_hw(x, fu, fd, cfg)]
    y = bias_act.bias_act(x=x, b=b)
    y = upfirdn2d.upfirdn2d(x=y, f=fu, up=up, padding=[cfg.px0, cfg.px1, cfg.py0, cfg.py1], gain=up ** 2, flip_filter=flip_filter)
    y = bias_act.bias_act(x=y, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
    y = upfirdn2d.upfirdn2d(x=y, f=fd, down=down, flip_filter=flip_filter)
    misc.assert_shape(y, want)
    assert y.dtype == x.dtype
    return y


def _as_2d_filter(f, factor, device):
    """None -> [[1]]; a one-tap separable filter without resampling -> its 2-D equivalent (the kernel wants 2-D there)."""
    if f is None:
        return torch.ones([1, 1], dtype=torch.float32, device=device)
    assert 1 <= f.ndim <= 2
    if factor == 1 and f.ndim == 1 and f.shape[0] == 1:
        return f.square()[None]
    return f


class _FilteredLReluHip(torch.autograd.Function):
    """Forward / backward on `ide3d_filtered_lrelu` (+ `ide3d_filtered_lrelu_act` for the generic path).  `signs` is the
    2-bit code tensor of an earlier forward pass (backward calls) or None; (sx, sy) its offset in the up-sampled frame."""

    @staticmethod
    def forward(ctx, x, fu, fd, b, signs, sx, sy, cfg):
        assert isinstance(x, torch.Tensor) and x.ndim == 4
        fu = _as_2d_filter(fu, cfg.up, x.device)
        fd = _as_2d_filter(fd, cfg.down, x.device)
        have_signs = signs is not None and signs.numel() > 0
        if not have_signs:
            signs = torch.empty([0])
        if b is None:
            b = torch.zeros([x.shape[1]], dtype=x.dtype, device=x.device)
        record = (not have_signs) and (x.requires_grad or b.requires_grad)
        live = [x.stride(i) for i in range(4) if x.size(i) > 1]
        if any(s0 < s1 for s0, s1 in zip(live, live[1:])):
            warnings.warn('low-performance memory layout detected in filtered_lrelu input', RuntimeWarning)

        pads = (cfg.px0, cfg.px1, cfg.py0, cfg.py1)
        rc = -1
        if x.dtype in (torch.float16, torch.float32, torch.bfloat16):
            # the HIP kernel keeps both filters in LDS (no global constant buffer): safe on concurrent streams
            y, codes, rc = _plugin.filtered_lrelu(x, fu, fd, b, signs, cfg.up, cfg.down, *pads, sx, sy,
                                                  cfg.gain, cfg.slope, cfg.clamp, cfg.flip, record)
        if rc < 0:
            warnings.warn('filtered_lrelu called with parameters that have no fused HIP kernel, using generic path', RuntimeWarning)
            y = x.add(b.reshape(1, -1, 1, 1))
            y = upfirdn2d.upfirdn2d(x=y, f=fu, up=cfg.up, padding=list(pads), gain=cfg.up ** 2, flip_filter=cfg.flip)
            codes = _plugin.filtered_lrelu_act_(y, signs, sx, sy, cfg.gain, cfg.slope, cfg.clamp, record)     # in place on y
            y = upfirdn2d.upfirdn2d(x=y, f=fd, down=cfg.down, flip_filter=cfg.flip)

        ctx.save_for_backward(fu, fd, signs if have_signs else codes)
        ctx.cfg, ctx.in_hw, ctx.out_hw, ctx.sign_ofs = cfg, tuple(x.shape[2:]), tuple(y.shape[2:]), (sx, sy)
        return y

    @staticmethod
    def backward(ctx, dy):
        fu, fd, codes = ctx.saved_tensors
        cfg = ctx.cfg
        need_x, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[3]
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2] or any(ctx.needs_input_grad[4:7]):
            raise NotImplementedError('filtered_lrelu: gradients w.r.t. filters / sign tensor are not defined')
        dx = db = None
        if need_x or need_b:
            (xh, xw), (yh, yw), (sx, sy) = ctx.in_hw, ctx.out_hw, ctx.sign_ofs
            taps_x = (fu.shape[-1] - 1) + (fd.shape[-1] - 1)
            taps_y = (fu.shape[0] - 1) + (fd.shape[0] - 1)
            # adjoint: resample factors and filters swap roles, padding is mirrored, activation becomes a per-element factor
            back = _Config(up=cfg.down, down=cfg.up,
                           px0=taps_x - cfg.px0, px1=xw * cfg.up - yw * cfg.down + cfg.px0 - (cfg.up - 1),
                           py0=taps_y - cfg.py0, py1=xh * cfg.up - yh * cfg.down + cfg.py0 - (cfg.up - 1),
                           gain=cfg.gain * cfg.up ** 2 / cfg.down ** 2, slope=cfg.slope, clamp=float('inf'), flip=not cfg.flip)
            dx = _FilteredLReluHip.apply(dy, fd, fu, None, codes, sx - (fu.shape[-1] - 1) + cfg.px0, sy - (fu.shape[0] - 1) + cfg.py0, back)
            if need_b:
                db = dx.sum([0, 2, 3])
        return dx, None, None, db, None, None, None, None
