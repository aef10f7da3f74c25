"""2-D convolution with optional up / down-sampling — the `torch_utils.ops.conv2d_resample.conv2d_resample` surface
(reference conv2d_resample.py:46).

Semantics: zero-insert upsample by `up`, pad, low-pass with `f` (gain up^2), convolve with `w`, low-pass again and keep
every `down`-th sample.  That definition is never executed literally; like the reference (:93-141) the work is ordered
so that the expensive convolution runs at the lowest possible resolution:
  * a 1x1 kernel commutes with the resampling filter (decimate first / interpolate last);
  * down-sampling = FIR at full resolution, then a stride-`down` convolution;
  * up-sampling   = stride-`up` transposed convolution, then the FIR with gain up^2.
Every `upfirdn2d` below lands on the HIP kernel for device tensors; the convolutions go through `conv2d_gradfix`.
"""

import torch

from .. import misc
from . import conv2d_gradfix
from . import upfirdn2d
from .upfirdn2d import _get_filter_size
from .upfirdn2d import _parse_padding


def _get_weight_shape(w):
    with misc.suppress_tracer_warnings():        # shapes as python ints even under tracing
        dims = [int(d) for d in w.shape]
    misc.assert_shape(w, dims)
    return dims


def _conv2d_wrapper(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """Correlation (`flip_weight=True`, what ATen computes) or true convolution (kernel mirrored first)."""
    kh, kw = _get_weight_shape(w)[2:]
    if not flip_weight and (kh > 1 or kw > 1):
        w = w.flip([2, 3])
    conv = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    return conv(x, w, stride=stride, padding=padding, groups=groups)


def _filter_padding(pads, f, up, down):
    """Padding requested by the caller plus the support the resampling filter needs (reference :73-78)."""
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = pads
    if up > 1:
        px0, px1 = px0 + (fw + up - 1) // 2, px1 + (fw - up) // 2
        py0, py1 = py0 + (fh + up - 1) // 2, py1 + (fh - up) // 2
    if down > 1:
        px0, px1 = px0 + (fw - down + 1) // 2, px1 + (fw - down) // 2
        py0, py1 = py0 + (fh - down + 1) // 2, py1 + (fh - down) // 2
    return px0, px1, py0, py1


def _transposed_weight(w, groups):
    """[O, I/g, kh, kw] -> the layout conv_transpose2d expects, per group."""
    o, ipg, kh, kw = _get_weight_shape(w)
    if groups == 1:
        return w.transpose(0, 1)
    w = w.reshape(groups, o // groups, ipg, kh, kw).transpose(1, 2)
    return w.reshape(groups * ipg, o // groups, kh, kw)


@misc.profiled_function
def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    """x [N, I, H, W], w [O, I/groups, kh, kw], f FIR (None / 1-D / 2-D, float32).  Arguments as in the reference."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in (1, 2) and f.dtype == torch.float32)
    for name, v in (('up', up), ('down', down), ('groups', groups)):
        assert isinstance(v, int) and v >= 1, f'{name} must be a positive int'
    _o, _ipg, kh, kw = _get_weight_shape(w)
    px0, px1, py0, py1 = _filter_padding(_parse_padding(padding), f, up, down)
    fir = dict(f=f, flip_filter=flip_filter)
    conv = dict(w=w, groups=groups, flip_weight=flip_weight)
    pointwise = kh == 1 and kw == 1

    if pointwise and up == 1 and down > 1:
        x = upfirdn2d.upfirdn2d(x=x, down=down, padding=[px0, px1, py0, py1], **fir)
        return _conv2d_wrapper(x=x, **conv)

    if pointwise and up > 1 and down == 1:
        x = _conv2d_wrapper(x=x, **conv)
        return upfirdn2d.upfirdn2d(x=x, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, **fir)

    if up == 1 and down > 1:
        x = upfirdn2d.upfirdn2d(x=x, padding=[px0, px1, py0, py1], **fir)
        return _conv2d_wrapper(x=x, stride=down, **conv)

    if up > 1:
        # the transposed conv itself provides kw - 1 / kw - up samples of "padding"; what is still missing (or in
        # excess: negative = crop) goes to the FIR, after moving as much cropping as possible into the conv
        px0, px1 = px0 - (kw - 1), px1 - (kw - up)
        py0, py1 = py0 - (kh - 1), py1 - (kh - up)
        cx, cy = max(min(-px0, -px1), 0), max(min(-py0, -py1), 0)
        x = _conv2d_wrapper(x=x, w=_transposed_weight(w, groups), stride=up, padding=[cy, cx], groups=groups, transpose=True,
                            flip_weight=not flip_weight)
        x = upfirdn2d.upfirdn2d(x=x, padding=[px0 + cx, px1 + cx, py0 + cy, py1 + cy], gain=up ** 2, **fir)
        return upfirdn2d.upfirdn2d(x=x, down=down, **fir) if down > 1 else x

    if px0 == px1 and py0 == py1 and min(px0, py0) >= 0:      # up == down == 1, symmetric padding: one plain convolution
        return _conv2d_wrapper(x=x, padding=[py0, px0], **conv)

    # asymmetric / negative padding without resampling: pad or crop explicitly
    x = upfirdn2d.upfirdn2d(x=x, f=None, up=1, padding=[px0, px1, py0, py1], gain=1, flip_filter=flip_filter)
    return _conv2d_wrapper(x=x, **conv)
