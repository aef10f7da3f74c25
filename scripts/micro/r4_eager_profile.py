"""Round 4: where the host time of the drop-in loop (batch 1, eager launches) goes.  cProfile over 60 images + GPU time of the same loop."""
import cProfile, pstats, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from training import triplane, distributed_render as dr
from torch_utils import hip_plugin
dev = torch.device('cuda:0')
torch.manual_seed(0)
G = triplane.TriPlaneGenerator().eval().to(dev)
cond = triplane.conditioning_label(dev); cam = triplane.camera_label(0.0, device=dev); pal = dr.palette_tensor(19, dev)

def one(seed):
    z = torch.from_numpy(np.random.RandomState(seed).randn(1, G.z_dim)).to(dev).float()
    with torch.no_grad():
        ws = G.mapping(z, cond)
        img, seg = G.synthesis(ws, c=cam, noise_mode='const', return_seg=True)
        return dr.frames_u8(img, seg, pal)

for s in range(5): one(s)
torch.cuda.synchronize()
calls0 = dict(hip_plugin.CALLS)
t0 = time.perf_counter()
for s in range(40): one(100 + s)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'per image: host enqueue {t_host / 40 * 1e3:.2f} ms, wall {t_all / 40 * 1e3:.2f} ms; native launches per image:',
      {k: (v - calls0.get(k, 0)) / 40 for k, v in hip_plugin.CALLS.items() if v != calls0.get(k, 0)})
# GPU time alone: the same image through a captured graph of batch 1
run = triplane.GraphedRenderer(G, 1, dev)
z = torch.randn(1, G.z_dim, device=dev)
for _ in range(5): run(z, cond, cam)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): run(z, cond, cam)
torch.cuda.synchronize()
print(f'batch-1 hipGraph replay: {(time.perf_counter() - t0) / 40 * 1e3:.2f} ms per image')
pr = cProfile.Profile(); pr.enable()
for s in range(40): one(200 + s)
pr.disable(); torch.cuda.synchronize()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(28); print(st.getvalue()[:6000])
