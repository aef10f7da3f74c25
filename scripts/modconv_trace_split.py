"""Cycles wave 0 of one workgroup of the split-bf16 convolution spends per phase of a stage (= 16 input channels x one kernel row;
library built with `make EXTRA=-DIDE3D_MC_TRACE`).  usage: python scripts/modconv_trace_split.py [conv|tconv] cin cout res arith"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd'))
import torch
from torch_utils import hip_plugin
kind, cin, cout, res, arith = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
n = 4
x = torch.randn(n, cin, res, res, generator=g).to(dev); w = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
s = (torch.randn(n, cin, generator=g) + 1).to(dev); d = torch.rand(n, cout, generator=g).to(dev)
f = lambda: hip_plugin.ModconvPlugin.modconv2d(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=(2 if kind == 'tconv' else 0), arith=arith)
for _ in range(3): f()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
lib = hip_plugin.load()
lib.ide3d_debug_mc.argtypes = [ctypes.c_void_p]
assert lib.ide3d_debug_mc(buf) == 0
names = ('loop head', 'staging issue (DMA + patch loads)', 'operand reads + MFMA', 'commit', 'vmcnt(0)', 'barrier')
for o, wave in ((0, 'wave 0'), (16, 'wave 4')):
    st = int(buf[o + 7])
    if not st:
        continue
    tot = sum(buf[o + k] for k in range(6))
    print(f'{kind} {cin}->{cout} @{res} arith {arith} {wave}: {st} stages, {tot / st:.0f} cycles per stage: ' +
          ', '.join(f'{nm} {buf[o + k] / st:.0f}' for k, nm in enumerate(names)) + f'; prologue {buf[o + 8]}, epilogue issue {buf[o + 9]}')
