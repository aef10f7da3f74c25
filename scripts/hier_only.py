"""Time the renderer with and without the importance pass (batch 4, 64 x 64 rays, 96 (+96) samples): HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd'))
import torch
from training import triplane

dev = torch.device('cuda:0')
torch.manual_seed(0)
R = triplane.TriplaneRenderer(triplane.GeneratorSpec()).to(dev).eval()
g = torch.Generator().manual_seed(1)
n = 4
tex = (torch.randn(n, 96, 256, 256, generator=g) * 0.7).to(dev).contiguous(memory_format=torch.channels_last)
geo = (torch.randn(n, 96, 256, 256, generator=g) * 0.7).to(dev).contiguous(memory_format=torch.channels_last)
cam = torch.cat([triplane.camera_label(y) for y in (-0.5, -0.15, 0.2, 0.5)])[:, :16].reshape(-1, 4, 4).to(dev)
for hier in (False, True):
    with torch.no_grad():
        for _ in range(3):
            R(tex, geo, cam, hierarchical=hier)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            R(tex, geo, cam, hierarchical=hier)
        b.record(); torch.cuda.synchronize()
    print(f'hierarchical={hier}: {a.elapsed_time(b) / 10:.3f} ms per batch of {n}')
