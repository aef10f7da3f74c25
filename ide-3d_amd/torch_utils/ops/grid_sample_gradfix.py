"""`grid_sample(bilinear, zeros, align_corners=False)` with optional second-order gradients
(`torch_utils.ops.grid_sample_gradfix` surface, reference grid_sample_gradfix.py:26).

The reference reaches the backward kernel through `torch._C._jit_get_operation`, which returns a
tuple on current PyTorch and fails; `torch.ops.aten.grid_sampler_2d_backward` is the
supported spelling.  The module-level switch `enabled` keeps its meaning (off by default).
"""

import torch

enabled = False  # set True to get arbitrary-order gradients


def grid_sample(input, grid):
    if _should_use_custom_op():
        return _GridSample2dForward.apply(input, grid)
    return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)


def _should_use_custom_op():
    return enabled


class _GridSample2dForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid):
        assert input.ndim == 4 and grid.ndim == 4
        ctx.save_for_backward(input, grid)
        return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        return _GridSample2dBackward.apply(grad_output, input, grid)


class _GridSample2dBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad_output, input, grid):
        grad_input, grad_grid = torch.ops.aten.grid_sampler_2d_backward(
            grad_output, input, grid, 0, 0, False, [True, True])
        ctx.save_for_backward(grid)
        return grad_input, grad_grid

    @staticmethod
    def backward(ctx, grad2_grad_input, grad2_grad_grid):
        grid, = ctx.saved_tensors
        grad2_grad_output = None
        if ctx.needs_input_grad[0]:
            grad2_grad_output = _GridSample2dForward.apply(grad2_grad_input, grid)
        assert not ctx.needs_input_grad[2]
        return grad2_grad_output, None, None
