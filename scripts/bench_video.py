"""BASELINE config 3: gen_videos.py 2x2 grid, 120-frame yaw / pitch sweep, image_seg dual-branch frames, one GPU.
Prints one JSON line: grid frames/s and 512x512 RGB+seg cell images/s of the FRAME LOOP (frames 2 .. n; `G.synthesis` replays its captured
hipGraph), the set-up before the first frame (mapping, scipy splines, upload) separately, and the whole job including it — with the
pose-independent tri-planes cached (4 static seeds) and without (what a latent interpolation needs)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd')); sys.path.insert(0, ROOT)
import torch
from training import triplane, video_render
from torch_utils import hip_plugin

frames_n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device('cuda:0')
torch.manual_seed(0)
G = triplane.TriPlaneGenerator().eval().to(dev)
res = {}
for name, cache in (('cached_triplanes', True), ('full_synthesis_every_frame', False)):
    for _ in video_render.gen_interp_frames(G, [0, 1, 2, 3], w_frames=4, grid_dims=(2, 2), device=dev, cache_static_planes=cache):
        pass                                                            # warm-up (weight packing, plugin init)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    gen = video_render.gen_interp_frames(G, [0, 1, 2, 3], w_frames=frames_n, grid_dims=(2, 2), device=dev, cache_static_planes=cache)
    frame = next(gen); n = 1                  # the job's set-up runs before the first frame: latents, mapping, the scipy splines of gen_videos.py:95-104, upload (+ tri-planes)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for frame in gen:
        n += 1
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res[name] = dict(grid_frames_per_s=(n - 1) / (t2 - t1), cell_images_per_s=4 * (n - 1) / (t2 - t1), ms_per_grid_frame=(t2 - t1) / (n - 1) * 1e3,
                     setup_and_first_frame_s=t1 - t0, whole_job_grid_frames_per_s=n / (t2 - t0), whole_job_cell_images_per_s=4 * n / (t2 - t0))
print(json.dumps(dict(metric='gen_videos 2x2 grid sweep, image_seg', frames=frames_n, grid_frame_shape=list(frame.shape), **res,
                      native_launches=dict(hip_plugin.CALLS))))
