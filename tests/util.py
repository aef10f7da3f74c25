"""Shared helpers for the test-suite."""

import numpy as np
import torch


def t(a, device='cpu', dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        x = x.to(dtype)
    return x.to(device)


def assert_close(actual, expected, rtol=1e-5, atol=1e-6, what=''):
    a = actual.detach().cpu().double().numpy() if isinstance(actual, torch.Tensor) else np.asarray(actual, dtype=np.float64)
    e = expected.detach().cpu().double().numpy() if isinstance(expected, torch.Tensor) else np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, f'{what}: shape {a.shape} != {e.shape}'
    err = np.abs(a - e)
    tol = atol + rtol * np.abs(e)
    if not np.all(err <= tol):
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f'{what}: max violation at {i}: got {a[i]!r}, expected {e[i]!r}, |err|={err[i]:.3e}, tol={tol[i]:.3e}; '
                             f'max abs err {err.max():.3e}')


def filter_from(arr):
    return None if arr.size == 0 else torch.from_numpy(np.ascontiguousarray(arr))
