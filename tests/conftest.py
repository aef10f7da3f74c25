"""pytest configuration: the `gpu` marker, import paths and the golden-fixture loader."""

import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OVERLAY = os.path.join(ROOT, 'ide-3d_amd')          # product: mirrors torch_utils / training / dnnlib of the reference
for p in (OVERLAY, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class Golden:
    """tests/golden/<name>.npz written by oracle/make_golden.py from the reference's own code."""

    def __init__(self, name):
        data = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
        self.cfgs = json.loads(bytes(data['cfg']).decode())
        self.cases = []
        for i, cfg in enumerate(self.cfgs):
            arrays = {k.split('/', 1)[1]: data[k] for k in data.files if k.startswith(f'{i}/')}
            self.cases.append((cfg, arrays))

    def __iter__(self):
        return iter(self.cases)

    def select(self, **match):
        return [(c, a) for c, a in self.cases if all(c.get(k) == v for k, v in match.items())]


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return load


@pytest.fixture(scope='session')
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    from torch_utils import custom_ops, hip_plugin
    if not os.path.isfile(hip_plugin.lib_path()):      # normally built by __graft_entry__.build() and shipped in-tree
        custom_ops.build_library(verbose=False)
    return torch.device('cuda:0')
