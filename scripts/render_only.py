"""Time the fused ray-marcher alone (benchmark shape: 4 images x 64x64 rays x 96 steps, 2 tri-planes) — target for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd')); sys.path.insert(0, ROOT)
import torch
from training import triplane
dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
R = triplane.TriplaneRenderer(triplane.GeneratorSpec()).to(dev).eval()
g = torch.Generator().manual_seed(1)
n = 4
tex = (torch.randn(n, 96, 256, 256, generator=g) * 0.7).to(dev).contiguous(memory_format=torch.channels_last)
geo = (torch.randn(n, 96, 256, 256, generator=g) * 0.7).to(dev).contiguous(memory_format=torch.channels_last)
cam = torch.cat([triplane.camera_label(y) for y in (-0.5, 0.0, 0.5, 0.25)])[:, :16].reshape(-1, 4, 4).to(dev)
jit = torch.rand(n, 4096, 96, generator=g).to(dev)
with torch.no_grad():
    for _ in range(3):
        R(tex, geo, cam, jitter=jit)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        R(tex, geo, cam, jitter=jit)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
samples = n * 4096 * 96
print(f'render_rays: {ms * 1e3:.1f} us per batch of {n}; {samples / ms / 1e6:.2f} Gsamples/s; MLP {samples * 2 * (32 * 64 * 2 + 64 * 52) / ms / 1e9:.1f} TFLOP/s')
