"""`conv2d` / `conv_transpose2d` whose gradients can be differentiated again — the `torch_utils.ops.conv2d_gradfix`
surface (reference conv2d_gradfix.py:21-44): module switches `enabled` and `weight_gradients_disabled`, the context
manager `no_weight_gradients()`, and the two functional entry points with `torch.nn.functional` signatures.

With `enabled = False` (default) both functions are the plain ATen convolutions (MIOpen / hipBLASLt on ROCm).  The hot
3x3 / 1x1 convolutions of the generator never come through here on device tensors in inference: `training.networks`
sends them to the fp32-MFMA implicit-GEMM kernel (`csrc/modconv.hip`).

With the switch on, a convolution is recorded as a `_Conv` node whose backward consists of further `_Conv` /
`_ConvWeightGrad` nodes, so gradient penalties (R1) can differentiate through it; weight gradients can be suppressed
for the regularisation passes.  The weight gradient comes from the dispatcher op `aten::convolution_backward` — the
reference's `torch._C._jit_get_operation('aten::cudnn_convolution_backward_weight')` no longer exists.
"""

import collections
import contextlib

import torch
import torch.nn.functional as F

enabled = False                      # True: build the double-differentiable graph for device tensors
weight_gradients_disabled = False    # True: `_Conv.backward` returns no weight gradient

# geometry of one convolution; `transpose` selects conv_transpose2d
_Geom = collections.namedtuple('_Geom', 'transpose stride padding output_padding dilation groups')


@contextlib.contextmanager
def no_weight_gradients(disable=True):
    """Inside the block (when `disable`) convolutions do not produce weight gradients."""
    global weight_gradients_disabled
    saved = weight_gradients_disabled
    weight_gradients_disabled = saved or bool(disable)
    try:
        yield
    finally:
        weight_gradients_disabled = saved


def _two(v):
    pair = tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    if len(pair) != 2 or not all(isinstance(e, int) for e in pair):
        raise AssertionError('convolution geometry arguments must be ints or pairs of ints')
    return pair


def _geometry(transpose, stride, padding, output_padding, dilation, groups):
    g = _Geom(bool(transpose), _two(stride), _two(padding), _two(output_padding), _two(dilation), int(groups))
    ok = g.groups >= 1 and min(g.stride) >= 1 and min(g.padding) >= 0 and min(g.dilation) >= 0
    if g.transpose:
        ok = ok and all(0 <= op < max(s, d) for op, s, d in zip(g.output_padding, g.stride, g.dilation))
    else:
        ok = ok and g.output_padding == (0, 0)
    if not ok:
        raise AssertionError(f'invalid convolution geometry {g}')
    return g


def _custom_graph_wanted(input):
    assert isinstance(input, torch.Tensor)
    return enabled and torch.backends.cudnn.enabled and input.device.type == 'cuda'


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if _custom_graph_wanted(input):
        return _Conv.apply(input, weight, bias, _geometry(False, stride, padding, 0, dilation, groups))
    return F.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding, dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    if _custom_graph_wanted(input):
        return _Conv.apply(input, weight, bias, _geometry(True, stride, padding, output_padding, dilation, groups))
    return F.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                              output_padding=output_padding, groups=groups, dilation=dilation)


def _run(geom, x, w, b=None):
    if geom.transpose:
        return F.conv_transpose2d(x, w, b, stride=geom.stride, padding=geom.padding, output_padding=geom.output_padding,
                                  groups=geom.groups, dilation=geom.dilation)
    return F.conv2d(x, w, b, stride=geom.stride, padding=geom.padding, dilation=geom.dilation, groups=geom.groups)


def _adjoint(geom, in_shape, out_shape, w_shape):
    """Geometry of the convolution that maps d(out) back to d(in): the opposite kind, with the output padding that makes
    the sizes match (a strided conv loses up to stride-1 rows / columns that its adjoint has to restore)."""
    if geom.transpose:
        out_pad = (0, 0)
    else:
        out_pad = tuple(in_shape[i + 2] - (out_shape[i + 2] - 1) * geom.stride[i] - (1 - 2 * geom.padding[i])
                        - geom.dilation[i] * (w_shape[i + 2] - 1) for i in range(2))
    return geom._replace(transpose=not geom.transpose, output_padding=out_pad)


_EMPTY = torch.empty([0])


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, geom):
        # keep only what some gradient will need
        ctx.save_for_backward(x if w.requires_grad else _EMPTY, w if x.requires_grad else _EMPTY)
        ctx.geom, ctx.x_shape, ctx.w_shape = geom, tuple(x.shape), tuple(w.shape)
        return _run(geom, x, w, b)

    @staticmethod
    def backward(ctx, d_out):
        x, w = ctx.saved_tensors
        geom = ctx.geom
        d_x = d_w = d_b = None
        if ctx.needs_input_grad[0]:
            d_x = _Conv.apply(d_out, w, None, _adjoint(geom, ctx.x_shape, tuple(d_out.shape), ctx.w_shape))
            assert tuple(d_x.shape) == ctx.x_shape
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            d_w = _ConvWeightGrad.apply(d_out, x, geom, ctx.w_shape)
        if ctx.needs_input_grad[2]:
            d_b = d_out.sum([0, 2, 3])
        return d_x, d_w, d_b, None


class _ConvWeightGrad(torch.autograd.Function):
    """d(loss)/d(weight) = correlation of d(out) with the input; linear in both, so its own backward is two more convs."""

    @staticmethod
    def forward(ctx, d_out, x, geom, w_shape):
        ctx.save_for_backward(d_out if x.requires_grad else _EMPTY, x if d_out.requires_grad else _EMPTY)
        ctx.geom, ctx.w_shape, ctx.x_shape, ctx.o_shape = geom, w_shape, tuple(x.shape), tuple(d_out.shape)
        shape_only = torch.empty(w_shape, dtype=x.dtype, device=x.device)
        _, d_w, _ = torch.ops.aten.convolution_backward(
            d_out, x, shape_only, None, list(geom.stride), list(geom.padding), list(geom.dilation),
            geom.transpose, list(geom.output_padding), geom.groups, [False, True, False])
        assert tuple(d_w.shape) == tuple(w_shape)
        return d_w

    @staticmethod
    def backward(ctx, dd_w):
        d_out, x = ctx.saved_tensors
        geom = ctx.geom
        dd_out = d_x = None
        if ctx.needs_input_grad[0]:
            dd_out = _Conv.apply(x, dd_w, None, geom)
            assert tuple(dd_out.shape) == ctx.o_shape
        if ctx.needs_input_grad[1]:
            d_x = _Conv.apply(d_out, dd_w, None, _adjoint(geom, ctx.x_shape, ctx.o_shape, ctx.w_shape))
            assert tuple(d_x.shape) == ctx.x_shape
        return dd_out, d_x, None, None
