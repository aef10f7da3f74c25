"""Cycles one wave of the convolution kernel spends per phase of its K loop (library built with
`make EXTRA=-DIDE3D_MC_TRACE`).  usage: python scripts/modconv_trace.py [tconv|conv|heads] cin cout res [images]"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd'))
import torch
from torch_utils import hip_plugin
kind, cin, cout, res = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
n = 4
x = torch.randn(n, cin, res, res, generator=g).to(dev); w = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
s = (torch.randn(n, cin, generator=g) + 1).to(dev); d = torch.rand(n, cout, generator=g).to(dev)
if kind == 'heads':          # per-image 1x1 weights (the folded RGB + seg heads), bias, clamp
    w = torch.randn(n, cout, cin, 1, 1, generator=g).to(dev); b = torch.randn(cout, generator=g).to(dev)
    f = lambda: hip_plugin.ModconvPlugin.modconv2d(x, w, None, None, None, 0.0, b, 1, 0.0, 1.0, 256.0)
elif kind == 'tconv':
    f = lambda: hip_plugin.ModconvPlugin.modconv2d(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=2)
else:
    f = lambda: hip_plugin.ModconvPlugin.modconv2d(x, w, s, d, None, 0.0, None, 3, 0.2, math.sqrt(2), -1.0)
for _ in range(3): f()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
lib = hip_plugin.load()
lib.ide3d_debug_mc.argtypes = [ctypes.c_void_p]
assert lib.ide3d_debug_mc(buf) == 0
chunks = max(int(buf[7]), 1)
names = ('loop', 'patch loads issue', 'operand reads + MFMA', 'commit + vmcnt(0)', 'barrier', 'weight DMA issue')
print(f'{kind} {cin}->{cout} @{res}: {chunks} chunks; cycles per chunk: ' +
      ', '.join(f'{nm} {buf[k] / chunks:.0f}' for k, nm in enumerate(names)) + f'; prologue (plan + first chunk) {buf[8]}, epilogue issue {buf[9]}')
