"""Host time of each part of the gen_images.py loop body (batch 1, library-captured graph): where an image's 1.8 ms go."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from training import triplane, graph_cache, distributed_render as dr
from training.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
from torch_utils import hip_plugin
dev = torch.device('cuda:0'); hip_plugin.load(); torch.manual_seed(0)
G = triplane.TriPlaneGenerator().eval().requires_grad_(False).to(dev)
cs = triplane.conditioning_label(dev); pal = dr.palette_tensor(19, dev)
T = {k: 0.0 for k in ('latent+mapping', 'pose', 'intrinsics .to(c) + cat', 'synthesis call', 'frames_u8')}
def tick(k, t0):
    t1 = time.perf_counter(); T[k] += t1 - t0; return t1
def seed_images(seed, acc):
    t = time.perf_counter()
    torch.manual_seed(seed)
    z = torch.from_numpy(np.random.RandomState(seed).randn(1, G.z_dim)).to(dev)
    ws = G.mapping(z=z, c=cs, truncation_psi=1)
    if acc: t = tick('latent+mapping', t)
    for yaw in (-0.5, 0, 0.5):
        t = time.perf_counter()
        rp = {'h_mean': yaw + math.pi * 0.5, 'v_mean': math.pi * 0.5, 'h_stddev': 0., 'v_stddev': 0., 'fov': 18, 'num_steps': 96}
        camera_points, phi, theta = sample_camera_positions(dev, n=1, r=2.7, horizontal_mean=yaw + math.pi * 0.5, vertical_mean=math.pi * 0.5, mode=None)
        c = create_cam2world_matrix(-camera_points, camera_points, device=dev)
        c = c.reshape(1, -1)
        if acc: t = tick('pose', t)
        c = torch.cat((c, torch.tensor([4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).reshape(1, -1).to(c)), -1)
        if acc: t = tick('intrinsics .to(c) + cat', t)
        img, seg = G.synthesis(ws, c=c, render_params=rp, noise_mode='const', return_seg=True)
        if acc: t = tick('synthesis call', t)
        dr.frames_u8(img, seg, pal)
        if acc: t = tick('frames_u8', t)
with torch.no_grad():
    for s in range(3): seed_images(1000 + s, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for s in range(n): seed_images(s, True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(json.dumps({'ms_per_image': round(dt / (3 * n) * 1e3, 3), 'host_us_per_image': {k: round(v / (3 * n) * 1e6, 1) for k, v in T.items()}}))
