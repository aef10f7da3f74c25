"""Fused bias + activation (`torch_utils.ops.bias_act` surface of the reference, bias_act.py:52).

CUDA(-device) tensors with `impl='cuda'` run the hand-written gfx950 kernel (`csrc/bias_act.hip`)
through the C ABI; CPU tensors or `impl='ref'` run the plain-PyTorch definition below, exactly as
the reference dispatches (bias_act.py:84-86).  A missing / broken HIP library is a hard error for
device tensors — there is no silent fallback.
"""

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
        return _bias_act_cuda(dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp).apply(x, b)
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


_bias_act_cuda_cache = dict()


def _bias_act_cuda(dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """autograd.Function bound to one (dim, act, alpha, gain, clamp) tuple; cached like bias_act.py:124-139."""
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    key = (dim, act, alpha, gain, clamp)
    if key in _bias_act_cuda_cache:
        return _bias_act_cuda_cache[key]

    trivial = (act == 'linear' and gain == 1 and clamp < 0)
    keep_x = ('x' in spec.ref) or spec.has_2nd_grad
    keep_y = 'y' in spec.ref

    def layout_of(t):
        return torch.channels_last if (t.ndim > 2 and t.stride(1) == 1) else torch.contiguous_format

    def null_like(_t):
        return _null_tensor

    class BiasActCuda(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, b):
            ctx.memory_format = layout_of(x)
            x = x.contiguous(memory_format=ctx.memory_format)
            b = b.contiguous() if b is not None else null_like(x)
            y = x
            if not trivial or b.numel():
                nul = null_like(x)
                y = _plugin.bias_act(x, b, nul, nul, nul, 0, dim, spec.cuda_idx, alpha, gain, clamp)
            ctx.save_for_backward(x if keep_x else null_like(x), b if keep_x else null_like(x), y if keep_y else null_like(x))
            return y

        @staticmethod
        def backward(ctx, dy):
            dy = dy.contiguous(memory_format=ctx.memory_format)
            x, b, y = ctx.saved_tensors
            dx = db = None
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                dx = dy if trivial else BiasActCudaGrad.apply(dy, x, b, y)
            if ctx.needs_input_grad[1]:
                db = dx.sum([i for i in range(dx.ndim) if i != dim])
            return dx, db

    class BiasActCudaGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, dy, x, b, y):
            ctx.memory_format = layout_of(dy)
            dx = _plugin.bias_act(dy, b, x, y, null_like(dy), 1, dim, spec.cuda_idx, alpha, gain, clamp)
            ctx.save_for_backward(dy if spec.has_2nd_grad else null_like(dy), x, b, y)
            return dx

        @staticmethod
        def backward(ctx, d_dx):
            d_dx = d_dx.contiguous(memory_format=ctx.memory_format)
            dy, x, b, y = ctx.saved_tensors
            d_dy = d_x = d_b = None
            if ctx.needs_input_grad[0]:
                d_dy = BiasActCudaGrad.apply(d_dx, x, b, y)
            if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                d_x = _plugin.bias_act(d_dx, b, x, y, dy, 2, dim, spec.cuda_idx, alpha, gain, clamp)
            if spec.has_2nd_grad and ctx.needs_input_grad[2]:
                d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
            return d_dy, d_x, d_b, None

    _bias_act_cuda_cache[key] = BiasActCuda
    return BiasActCuda
