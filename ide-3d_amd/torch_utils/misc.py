"""Small tensor helpers used by the ops (subset of the reference's torch_utils/misc.py)."""

import contextlib
import warnings

import torch


@contextlib.contextmanager
def suppress_tracer_warnings():
    """Silence torch.jit.TracerWarning inside the block (reference: misc.py:71)."""
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=torch.jit.TracerWarning)
        yield


def assert_shape(tensor, ref_shape):
    """Check `tensor.shape` against `ref_shape`; `None` entries are wildcards (reference: misc.py:82)."""
    if tensor.ndim != len(ref_shape):
        raise AssertionError(f'Wrong number of dimensions: got {tensor.ndim}, expected {len(ref_shape)}')
    for idx, (size, ref) in enumerate(zip(tensor.shape, ref_shape)):
        if ref is None:
            continue
        if isinstance(ref, torch.Tensor):
            with suppress_tracer_warnings():
                torch._assert(torch.equal(torch.as_tensor(size), ref), f'Wrong size for dimension {idx}')
        elif isinstance(size, torch.Tensor):
            with suppress_tracer_warnings():
                torch._assert(torch.equal(size, torch.as_tensor(ref)), f'Wrong size for dimension {idx}: expected {ref}')
        elif size != ref:
            raise AssertionError(f'Wrong size for dimension {idx}: got {size}, expected {ref}')


def profiled_function(fn):
    """Wrap `fn` in a torch profiler range named after it (reference: misc.py:100)."""
    def decorator(*args, **kwargs):
        with torch.autograd.profiler.record_function(fn.__name__):
            return fn(*args, **kwargs)
    decorator.__name__ = fn.__name__
    decorator.__doc__ = fn.__doc__
    return decorator


def named_params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.named_parameters()) + list(module.named_buffers())


def copy_params_and_buffers(src_module, dst_module, require_all=False):
    """Copy same-named tensors from src to dst (reference: misc.py:155)."""
    src = dict(named_params_and_buffers(src_module))
    for name, tensor in named_params_and_buffers(dst_module):
        assert (name in src) or (not require_all), f'missing tensor {name}'
        if name in src:
            tensor.copy_(src[name].detach()).requires_grad_(tensor.requires_grad)
