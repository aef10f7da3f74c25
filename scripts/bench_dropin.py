"""The reference's gen_images.py:88-114 loop shape on the overlay, loop body unchanged (batch 1: mapping, synthesis with `render_params`,
uint8 frame), timed; `G.synthesis` captures and replays its own hipGraph (training/graph_cache.py).
    python scripts/bench_dropin.py [images] [--eager]        ->  one JSON line
Under `rocprofv3 --kernel-trace` the frame_u8 launches delimit the images (scripts/step_timeline.py reads that trace)."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from training import triplane, graph_cache, distributed_render as dr
from training.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
from torch_utils import hip_plugin

images = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 48
eager = '--eager' in sys.argv
dev = torch.device('cuda:0')
hip_plugin.load()
torch.manual_seed(0)
G = triplane.TriPlaneGenerator().eval().requires_grad_(False).to(dev)          # G_ema of a released pickle does not require grad either
cs = triplane.conditioning_label(dev); pal = dr.palette_tensor(19, dev)


def seed_images(seed):
    torch.manual_seed(seed)
    z = torch.from_numpy(np.random.RandomState(seed).randn(1, G.z_dim)).to(dev)
    ws = G.mapping(z=z, c=cs, truncation_psi=1)
    out = []
    for yaw in (-0.5, 0, 0.5):
        rp = {'h_mean': yaw + math.pi * 0.5, 'v_mean': math.pi * 0.5, 'h_stddev': 0., 'v_stddev': 0., 'fov': 18, 'num_steps': 96}
        # gen_images.py:104-107, verbatim (the `.to(c)` of the intrinsics is the driver's own pageable upload: one synchronisation per pose)
        camera_points, phi, theta = sample_camera_positions(dev, n=1, r=2.7, horizontal_mean=yaw + math.pi * 0.5, vertical_mean=math.pi * 0.5, mode=None)
        c = create_cam2world_matrix(-camera_points, camera_points, device=dev)
        c = c.reshape(1, -1)
        c = torch.cat((c, torch.tensor([4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).reshape(1, -1).to(c)), -1)
        img, seg = G.synthesis(ws, c=c, render_params=rp, noise_mode='const', return_seg=True)
        out.append(dr.frames_u8(img, seg, pal))
    return out


ctx = graph_cache.disabled() if eager else torch.no_grad()
with ctx:
    for s in range(2):
        seed_images(1000 + s)
    torch.cuda.synchronize()
    before = dict(graph_cache.STATS)
    t0 = time.perf_counter()
    for s in range(images // 3):
        seed_images(s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
n = (images // 3) * 3
print(json.dumps({'what': 'gen_images.py:88-114 loop shape (1 seed -> mapping -> 3 yaws x synthesis + uint8 frame), batch 1, ' + ('eager launches' if eager else 'library-captured hipGraph'),
                  'images': n, 'frames_per_s': round(n / dt, 1), 'ms_per_image': round(dt / n * 1e3, 3), 'conv_arithmetic': hip_plugin.conv_arithmetic(),
                  'synthesis_calls': {k: graph_cache.STATS[k] - before.get(k, 0) for k in ('eager', 'capture', 'replay')}}))
