"""`conv2d` / `conv_transpose2d` with arbitrary-order gradients
(`torch_utils.ops.conv2d_gradfix` surface, reference conv2d_gradfix.py:35,40).

Forward and backward are ATen convolutions (MIOpen / hipBLASLt on ROCm).  The hot 3x3 / 1x1 modulated
convolutions of the generator do not come through here on device tensors: `training.networks`
routes them to the fp32-MFMA implicit-GEMM kernel (`csrc/modconv.hip`).  This module keeps the
reference's switches (`enabled`, `weight_gradients_disabled`, `no_weight_gradients`) and its
gradient structure, using `torch.ops.aten.convolution_backward` instead of the removed
`cudnn_convolution_backward_weight` private op (reference :171-173).
"""

import contextlib

import torch

enabled = False                     # enable the custom autograd graph
weight_gradients_disabled = False   # force-skip weight gradients


@contextlib.contextmanager
def no_weight_gradients(disable=True):
    global weight_gradients_disabled
    old = weight_gradients_disabled
    if disable:
        weight_gradients_disabled = True
    yield
    weight_gradients_disabled = old


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if _should_use_custom_op(input):
        return _conv2d_gradfix(transpose=False, weight_shape=weight.shape, stride=stride, padding=padding,
                               output_padding=0, dilation=dilation, groups=groups).apply(input, weight, bias)
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                      dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    if _should_use_custom_op(input):
        return _conv2d_gradfix(transpose=True, weight_shape=weight.shape, stride=stride, padding=padding,
                               output_padding=output_padding, groups=groups, dilation=dilation).apply(input, weight, bias)
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                                output_padding=output_padding, groups=groups, dilation=dilation)


def _should_use_custom_op(input):
    assert isinstance(input, torch.Tensor)
    if (not enabled) or (not torch.backends.cudnn.enabled):
        return False
    return input.device.type == 'cuda'


def _pair(v):
    v = tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    assert len(v) == 2 and all(isinstance(e, int) for e in v)
    return v


_conv2d_gradfix_cache = dict()
_null_tensor = torch.empty([0])


def _conv2d_gradfix(transpose, weight_shape, stride, padding, output_padding, dilation, groups):
    weight_shape = tuple(weight_shape)
    stride, padding, output_padding, dilation = _pair(stride), _pair(padding), _pair(output_padding), _pair(dilation)
    key = (transpose, weight_shape, stride, padding, output_padding, dilation, groups)
    if key in _conv2d_gradfix_cache:
        return _conv2d_gradfix_cache[key]

    assert groups >= 1 and len(weight_shape) == 4
    assert all(s >= 1 for s in stride) and all(p >= 0 for p in padding) and all(d >= 0 for d in dilation)
    if not transpose:
        assert output_padding == (0, 0)
    else:
        assert all(0 <= output_padding[i] < max(stride[i], dilation[i]) for i in range(2))

    kw = dict(stride=stride, padding=padding, dilation=dilation, groups=groups)

    def out_pad_for(input_shape, output_shape):
        if transpose:
            return [0, 0]
        return [input_shape[i + 2] - (output_shape[i + 2] - 1) * stride[i] - (1 - 2 * padding[i])
                - dilation[i] * (weight_shape[i + 2] - 1) for i in range(2)]

    class Conv2d(torch.autograd.Function):
        @staticmethod
        def forward(ctx, input, weight, bias):
            assert weight.shape == weight_shape
            ctx.save_for_backward(input if weight.requires_grad else _null_tensor,
                                  weight if input.requires_grad else _null_tensor)
            ctx.input_shape = input.shape
            if transpose:
                return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, output_padding=output_padding, **kw)
            return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, **kw)

        @staticmethod
        def backward(ctx, grad_output):
            input, weight = ctx.saved_tensors
            grad_input = grad_weight = grad_bias = None
            if ctx.needs_input_grad[0]:
                p = out_pad_for(ctx.input_shape, grad_output.shape)
                op = _conv2d_gradfix(transpose=(not transpose), weight_shape=weight_shape, output_padding=p, **kw)
                grad_input = op.apply(grad_output, weight, None)
                assert grad_input.shape == ctx.input_shape
            if ctx.needs_input_grad[1] and not weight_gradients_disabled:
                grad_weight = Conv2dGradWeight.apply(grad_output, input)
                assert grad_weight.shape == weight_shape
            if ctx.needs_input_grad[2]:
                grad_bias = grad_output.sum([0, 2, 3])
            return grad_input, grad_weight, grad_bias

    class Conv2dGradWeight(torch.autograd.Function):
        @staticmethod
        def forward(ctx, grad_output, input):
            ctx.save_for_backward(grad_output if input.requires_grad else _null_tensor,
                                  input if grad_output.requires_grad else _null_tensor)
            ctx.grad_output_shape = grad_output.shape
            ctx.input_shape = input.shape
            dummy_w = torch.empty(weight_shape, dtype=input.dtype, device=input.device)
            _, grad_weight, _ = torch.ops.aten.convolution_backward(
                grad_output, input, dummy_w, None, list(stride), list(padding), list(dilation),
                transpose, list(output_padding), groups, [False, True, False])
            return grad_weight

        @staticmethod
        def backward(ctx, grad2_grad_weight):
            grad_output, input = ctx.saved_tensors
            grad2_grad_output = grad2_input = None
            if ctx.needs_input_grad[0]:
                grad2_grad_output = Conv2d.apply(input, grad2_grad_weight, None)
                assert grad2_grad_output.shape == ctx.grad_output_shape
            if ctx.needs_input_grad[1]:
                p = out_pad_for(ctx.input_shape, ctx.grad_output_shape)
                op = _conv2d_gradfix(transpose=(not transpose), weight_shape=weight_shape, output_padding=p, **kw)
                grad2_input = op.apply(grad_output, grad2_grad_weight, None)
                assert grad2_input.shape == ctx.input_shape
            return grad2_grad_output, grad2_input

    _conv2d_gradfix_cache[key] = Conv2d
    return Conv2d
