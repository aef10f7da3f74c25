"""Round 5: where a full-synthesis frame of training/video_render.gen_interp_frames spends its time (host enqueue vs GPU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from training import triplane, video_render, graph_cache, distributed_render as dr
dev = torch.device('cuda:0')
torch.manual_seed(0)
G = triplane.TriPlaneGenerator().eval().to(dev)
lookat = torch.tensor([0, 0, 0.2], device=dev)
ws = torch.randn(4, G.num_ws, G.w_dim, device=dev)
pal = dr.palette_tensor(19, dev)

def t(fn, n=100):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return th / n * 1e3, (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    print('sweep_pose            host %.3f ms  wall %.3f ms' % t(lambda: video_render.sweep_pose(7, 120, lookat, device=dev).repeat(4, 1)))
    c = video_render.sweep_pose(7, 120, lookat, device=dev).repeat(4, 1)
    for _ in range(3): out = G.synthesis(ws, c=c, noise_mode='const', return_seg=True)
    print('synthesis (replay)    host %.3f ms  wall %.3f ms' % t(lambda: G.synthesis(ws, c=c, noise_mode='const', return_seg=True)), dict(graph_cache.STATS))
    img, seg = out
    print('frames_u8 + layout    host %.3f ms  wall %.3f ms' % t(lambda: video_render.layout_u8(dr.frames_u8(img, seg, pal), 2, 2)))
    def frame():
        c = video_render.sweep_pose(7, 120, lookat, device=dev).repeat(4, 1)
        img, seg = G.synthesis(ws, c=c, noise_mode='const', return_seg=True)
        return video_render.layout_u8(dr.frames_u8(img, seg, pal), 2, 2)
    print('whole frame           host %.3f ms  wall %.3f ms' % t(frame))
    run = triplane.GraphedRenderer(G, 4, dev)
    z = torch.randn(4, G.z_dim, device=dev); cond = triplane.conditioning_label(dev).repeat(4, 1)
    print('GraphedRenderer b4    host %.3f ms  wall %.3f ms' % t(lambda: run(z, cond, c)))
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): frame()
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22); print(s.getvalue()[:3500])
