"""Bilinear `grid_sample` (zeros padding, `align_corners=False`) whose input-gradient is itself differentiable.

Public surface of the reference module (torch_utils/ops/grid_sample_gradfix.py:22-31): the module switch `enabled`
(off by default) and `grid_sample(input, grid)`.  With the switch off this is exactly
`torch.nn.functional.grid_sample`.  With it on, sampling goes through a pair of autograd functions so that
d(output)/d(input) can be differentiated again (R1-style penalties on the sampled features): the gradient of a bilinear
look-up w.r.t. its input is linear in the incoming gradient, hence its own derivative w.r.t. that gradient is another
bilinear look-up at the same grid.

The reference obtains the backward kernel with `torch._C._jit_get_operation('aten::grid_sampler_2d_backward')`, which
returns a tuple on current PyTorch and fails when called; here the dispatcher entry
`torch.ops.aten.grid_sampler_2d_backward` (with its `output_mask`) is used.
"""

import torch
import torch.nn.functional as F

enabled = False     # True: route through the double-differentiable functions below

_BILINEAR, _ZEROS = 0, 0        # ATen interpolation / padding mode codes


def _sample(image, grid):
    return F.grid_sample(image, grid, mode='bilinear', padding_mode='zeros', align_corners=False)


def grid_sample(input, grid):
    """[N, C, H, W] sampled at `grid` [N, Ho, Wo, 2] (x, y in [-1, 1]) -> [N, C, Ho, Wo]."""
    return _Lookup.apply(input, grid) if enabled else _sample(input, grid)


class _Lookup(torch.autograd.Function):
    """out = sample(image, grid); backward defers to `_LookupGrad` so that it stays on the tape."""

    @staticmethod
    def forward(ctx, image, grid):
        if image.ndim != 4 or grid.ndim != 4:
            raise ValueError('grid_sample expects a 4-D image and a 4-D grid')
        ctx.save_for_backward(image, grid)
        return _sample(image, grid)

    @staticmethod
    def backward(ctx, d_out):
        image, grid = ctx.saved_tensors
        return _LookupGrad.apply(d_out, image, grid)


class _LookupGrad(torch.autograd.Function):
    """(d_image, d_grid) of the look-up.  Differentiable w.r.t. the incoming gradient only (never w.r.t. the grid:
    second derivatives through the coordinates are not needed by any caller and raise)."""

    @staticmethod
    def forward(ctx, d_out, image, grid):
        ctx.save_for_backward(grid)
        d_image, d_grid = torch.ops.aten.grid_sampler_2d_backward(d_out, image, grid, _BILINEAR, _ZEROS, False, [True, True])
        return d_image, d_grid

    @staticmethod
    def backward(ctx, dd_image, _dd_grid):
        if ctx.needs_input_grad[2]:
            raise NotImplementedError('second-order gradients w.r.t. the sampling grid are not supported')
        (grid,) = ctx.saved_tensors
        dd_out = _Lookup.apply(dd_image, grid) if ctx.needs_input_grad[0] else None
        return dd_out, None, None
