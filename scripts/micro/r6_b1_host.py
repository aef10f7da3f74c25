"""Host time of each part of bench.py's `dropin_b1` loop body (one seed per image): where an image's 1.68 ms go."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from training import triplane, graph_cache, distributed_render as dr
from torch_utils import hip_plugin
dev = torch.device('cuda:0'); hip_plugin.load(); torch.manual_seed(0)
G = triplane.TriPlaneGenerator().eval().requires_grad_(False).to(dev)
cond = triplane.conditioning_label(dev); cam = triplane.camera_label(0.0, device=dev); pal = dr.palette_tensor(19, dev)
T = {k: 0.0 for k in ('latents: from_numpy().to(device).float()', 'G.mapping', 'G.synthesis', 'frames_u8')}
def one(seed, acc):
    t0 = time.perf_counter()
    z = torch.from_numpy(np.random.RandomState(seed).randn(1, G.z_dim)).to(dev).float()
    t1 = time.perf_counter()
    with torch.no_grad():
        ws = G.mapping(z, cond)
        t2 = time.perf_counter()
        img, seg = G.synthesis(ws, c=cam, noise_mode='const', return_seg=True)
        t3 = time.perf_counter()
        dr.frames_u8(img, seg, pal)
    t4 = time.perf_counter()
    if acc:
        for k, d in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)): T[k] += d
for s in range(6): one(1000 + s, False)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 60
for s in range(n): one(s, True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
# the numpy draw alone (host only)
t0 = time.perf_counter()
for s in range(n): np.random.RandomState(s).randn(1, G.z_dim)
rs = (time.perf_counter() - t0) / n
print(json.dumps({'ms_per_image': round(dt / n * 1e3, 3), 'host_us_per_image': {k: round(v / n * 1e6, 1) for k, v in T.items()}, 'numpy_RandomState_randn_us': round(rs * 1e6, 1)}))
