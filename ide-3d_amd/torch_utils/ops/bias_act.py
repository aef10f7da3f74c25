"""Fused bias + activation (`torch_utils.ops.bias_act` surface of the reference, bias_act.py:52).

CUDA(-device) tensors with `impl='cuda'` run the hand-written gfx950 kernel (`csrc/bias_act.hip`)
through the C ABI; CPU tensors or `impl='ref'` run the plain-PyTorch definition below, exactly as
the reference dispatches (bias_act.py:84-86).  A missing / broken HIP library is a hard error for
device tensors — there is no silent fallback.
"""

import collections

import numpy as np
import torch

import dnnlib

from .. import custom_ops
from .. import misc

# name -> (python function, default alpha, default gain, kernel index, tensors needed by the gradient, 2nd-order?)
_ACT_TABLE = [
    ('linear',   lambda x, **_: x,                                         0.0, 1.0,          1, '',  False),
    ('relu',     lambda x, **_: torch.nn.functional.relu(x),               0.0, np.sqrt(2),   2, 'y', False),
    ('lrelu',    lambda x, alpha, **_: torch.nn.functional.leaky_relu(x, alpha), 0.2, np.sqrt(2), 3, 'y', False),
    ('tanh',     lambda x, **_: torch.tanh(x),                             0.0, 1.0,          4, 'y', True),
    ('sigmoid',  lambda x, **_: torch.sigmoid(x),                          0.0, 1.0,          5, 'y', True),
    ('elu',      lambda x, **_: torch.nn.functional.elu(x),                0.0, 1.0,          6, 'y', True),
    ('selu',     lambda x, **_: torch.nn.functional.selu(x),               0.0, 1.0,          7, 'y', True),
    ('softplus', lambda x, **_: torch.nn.functional.softplus(x),           0.0, 1.0,          8, 'y', True),
    ('swish',    lambda x, **_: torch.sigmoid(x) * x,                      0.0, np.sqrt(2),   9, 'x', True),
]

activation_funcs = {
    name: dnnlib.EasyDict(func=fn, def_alpha=a, def_gain=g, cuda_idx=idx, ref=ref, has_2nd_grad=second)
    for name, fn, a, g, idx, ref, second in _ACT_TABLE
}

_plugin = None
_null_tensor = torch.empty([0])


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(module_name='bias_act_plugin', sources=['bias_act.hip'], headers=['common.h'])
    return True


def _resolve(act, alpha, gain, clamp):
    assert clamp is None or clamp >= 0
    spec = activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    return spec, alpha, gain, clamp


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """y = clamp(act(x + b) * gain).  Same arguments and defaults as the reference (bias_act.py:52-81)."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
        return _BiasActHip.apply(x, b, _Call(dim, act, alpha, gain, clamp))
    return _bias_act_ref(x=x, b=b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)


@misc.profiled_function
def _bias_act_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """Plain-PyTorch definition (CPU path; reference bias_act.py:91)."""
    assert isinstance(x, torch.Tensor)
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1
        assert 0 <= dim < x.ndim
        assert b.shape[0] == x.shape[dim]
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    y = spec.func(x, alpha=alpha)
    if gain != 1:
        y = y * gain
    if clamp >= 0:
        y = y.clamp(-clamp, clamp)
    return y


# ---- device path: `ide3d_bias_act` evaluates the op (grad = 0), its derivative applied to dy (grad = 1) and the second
# derivative term (grad = 2); the autograd functions below only decide which tensors each order needs to keep. ----------

_Call = collections.namedtuple('_Call', 'dim act alpha gain clamp')      # resolved, hashable parameters of one call


def _layout(t):
    return torch.channels_last if (t.ndim > 2 and t.stride(1) == 1) else torch.contiguous_format


def _is_identity(call):
    return call.act == 'linear' and call.gain == 1 and call.clamp < 0


def _launch(call, grad, x, b, xref, yref, dy):
    return _plugin.bias_act(x, b, xref, yref, dy, grad, call.dim, activation_funcs[call.act].cuda_idx, call.alpha, call.gain, call.clamp)


def _sum_to_bias(t, dim):
    return t.sum([i for i in range(t.ndim) if i != dim])


class _BiasActHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, call):
        spec = activation_funcs[call.act]
        ctx.call, ctx.layout = call, _layout(x)
        x = x.contiguous(memory_format=ctx.layout)
        b = _null_tensor if b is None else b.contiguous()
        y = x if (_is_identity(call) and not b.numel()) else _launch(call, 0, x, b, _null_tensor, _null_tensor, _null_tensor)
        wants_x = ('x' in spec.ref) or spec.has_2nd_grad        # what the derivative is expressed in
        ctx.save_for_backward(x if wants_x else _null_tensor, b if wants_x else _null_tensor, y if 'y' in spec.ref else _null_tensor)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b, y = ctx.saved_tensors
        call = ctx.call
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dy = dy.contiguous(memory_format=ctx.layout)
            dx = dy if _is_identity(call) else _BiasActHipGrad.apply(dy, x, b, y, call)
            if ctx.needs_input_grad[1]:
                db = _sum_to_bias(dx, call.dim)
        return dx, db, None


class _BiasActHipGrad(torch.autograd.Function):
    """dx = dy * act'(.) * gain (zero where clamped); differentiable in dy always, in x / b for the smooth activations."""

    @staticmethod
    def forward(ctx, dy, x, b, y, call):
        ctx.call, ctx.layout = call, _layout(dy)
        smooth = activation_funcs[call.act].has_2nd_grad
        ctx.save_for_backward(dy if smooth else _null_tensor, x, b, y)
        return _launch(call, 1, dy, b, x, y, _null_tensor)

    @staticmethod
    def backward(ctx, d_dx):
        dy, x, b, y = ctx.saved_tensors
        call = ctx.call
        smooth = activation_funcs[call.act].has_2nd_grad
        d_dx = d_dx.contiguous(memory_format=ctx.layout)
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActHipGrad.apply(d_dx, x, b, y, call)
        if smooth and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _launch(call, 2, d_dx, b, x, y, dy)
            if ctx.needs_input_grad[2]:
                d_b = _sum_to_bias(d_x, call.dim)
        return d_dy, d_x, d_b, None, None
