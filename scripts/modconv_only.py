"""Time csrc/modconv.hip alone on the big layers of the generator (target for rocprofv3)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd')); sys.path.insert(0, ROOT)
import torch
from torch_utils import hip_plugin
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for (n, cin, cout, res, k) in ((4, 128, 128, 256, 3), (4, 512, 512, 64, 3), (4, 256, 256, 128, 3), (4, 64, 64, 512, 3), (4, 128, 96, 256, 1)):
    x = torch.randn(n, cin, res, res, generator=g).to(dev); w = torch.randn(cout, cin, k, k, generator=g).to(dev)
    s = (torch.randn(n, cin, generator=g) + 1).to(dev); d = torch.rand(n, cout, generator=g).to(dev)
    nz = torch.randn(res, res, generator=g).to(dev); b = torch.randn(cout, generator=g).to(dev)
    f = lambda: hip_plugin.ModconvPlugin.modconv2d(x, w, s, d, nz, 1.0, b, 3, 0.2, math.sqrt(2), -1.0)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2 * cin * cout * k * k * res * res * n
    print(f'modconv n={n} {cin}->{cout} @{res} k={k}: {ms*1e3:.1f} us, {fl/ms/1e9:.1f} TFLOP/s ({fl/ms/1e9/157.3*100:.1f}% of fp32 MFMA peak)')
print('--- transposed 3x3 stride 2 (mode 2) ---')
for (n, cin, cout, res) in ((4, 512, 256, 64), (4, 256, 128, 128), (4, 128, 64, 256), (4, 512, 512, 32)):
    x = torch.randn(n, cin, res, res, generator=g).to(dev); w = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
    s = (torch.randn(n, cin, generator=g) + 1).to(dev); d = torch.rand(n, cout, generator=g).to(dev)
    f = lambda: hip_plugin.ModconvPlugin.modconv2d(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=2)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2 * cin * cout * 9 * res * res * n
    print(f'tconv n={n} {cin}->{cout} in@{res}: {ms*1e3:.1f} us, {fl/ms/1e9:.1f} TFLOP/s ({fl/ms/1e9/157.3*100:.1f}% of fp32 MFMA peak)')
print('--- dual heads: per-image 1x1 weights, 96 + 96 (voxel blocks) and 3 + 19 (SR blocks) output channels ---')
for (n, cin, cout, res) in ((4, 128, 192, 256), (4, 256, 192, 128), (4, 512, 192, 64), (4, 64, 22, 512), (4, 128, 22, 256)):
    x = torch.randn(n, cin, res, res, generator=g).to(dev); w = torch.randn(n, cout, cin, 1, 1, generator=g).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    f = lambda: hip_plugin.ModconvPlugin.modconv2d(x, w, None, None, None, 0.0, b, 1, 0.0, 1.0, 256.0)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2 * cin * cout * res * res * n
    by = (cin + cout) * res * res * n * 4
    print(f'heads n={n} {cin}->{cout} @{res}: {ms*1e3:.1f} us, {fl/ms/1e9:.1f} TFLOP/s, {by/ms/1e6:.0f} GB/s of x + y')
