"""`fma(a, b, c) = a * b + c` with cheap gradients (`torch_utils.ops.fma` surface, reference fma.py:15)."""

import torch


def fma(a, b, c):
    return _FusedMultiplyAdd.apply(a, b, c)


class _FusedMultiplyAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da = _unbroadcast(dout * b, a.shape) if ctx.needs_input_grad[0] else None
        db = _unbroadcast(dout * a, b.shape) if ctx.needs_input_grad[1] else None
        dc = _unbroadcast(dout, ctx.c_shape) if ctx.needs_input_grad[2] else None
        return da, db, dc


def _unbroadcast(x, shape):
    """Sum `x` over the axes that were broadcast to reach its shape from `shape`."""
    lead = x.ndim - len(shape)
    assert lead >= 0
    axes = [i for i in range(x.ndim) if x.shape[i] > 1 and (i < lead or shape[i - lead] == 1)]
    if axes:
        x = x.sum(dim=axes, keepdim=True)
    if lead:
        x = x.reshape(-1, *x.shape[lead + 1:])
    assert x.shape == shape
    return x
