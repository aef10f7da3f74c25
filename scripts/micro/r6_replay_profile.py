"""cProfile of G.synthesis replays (batch 1): which host lines sit between the caller and the graph launch."""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
    sys.path.insert(0, p)
import torch
from training import triplane
from torch_utils import hip_plugin
dev = torch.device('cuda:0'); hip_plugin.load(); torch.manual_seed(0)
G = triplane.TriPlaneGenerator().eval().requires_grad_(False).to(dev)
cond = triplane.conditioning_label(dev); cam = triplane.camera_label(0.0, device=dev)
with torch.no_grad():
    ws = G.mapping(torch.randn(1, G.z_dim, device=dev), cond)
    for _ in range(5): G.synthesis(ws, c=cam, noise_mode='const', return_seg=True)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200):
        G.synthesis(ws, c=cam, noise_mode='const', return_seg=True)
        torch.cuda.synchronize()
    pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:6000])
