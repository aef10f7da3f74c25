"""Round 5 debugging aid: hipGraph capture / destroy churn through training/graph_cache.py (LRU bound 3, four signatures), optionally after
the things tests/test_gpu_graph_cache.py does before its soak test.  Prints progress so that a crash can be located."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from training import triplane, graph_cache
dev = torch.device('cuda:0')
mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
torch.manual_seed(3)
G = triplane.TriPlaneGenerator(triplane.tiny_spec()).eval().requires_grad_(False).to(dev)
def ws_of(G, seeds):
    z = torch.from_numpy(np.stack([np.random.RandomState(s).randn(G.z_dim) for s in seeds])).to(dev)
    return G.mapping(z, triplane.conditioning_label(dev).repeat(len(seeds), 1))
cam = lambda ys: torch.cat([triplane.camera_label(y, device=dev) for y in ys])
if 'full' in mode:
    torch.manual_seed(0)
    GF = triplane.TriPlaneGenerator().eval().requires_grad_(False).to(dev)
    w = ws_of(GF, [0])
    for _ in range(4):
        GF.synthesis(w, c=cam([0.0]), noise_mode='const', return_seg=True)
    print('full-size graph alive:', graph_cache.stats(GF.synthesis), flush=True)
if 'threads' in mode:
    c = cam([0.1]); wss = [ws_of(G, [20 + k]) for k in range(2)]
    def worker(k):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st), torch.no_grad():
            for _ in range(6):
                G.synthesis(wss[k], c=c, ray_jitter=False, return_seg=True)
            st.synchronize()
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    print('threads done:', graph_cache.stats(G.synthesis), flush=True)
graph_cache.reset(G.synthesis)
os.environ['IDE3D_AUTO_GRAPH_MAX'] = '3'
c = cam([0.0])
wss = {n: ws_of(G, list(range(n))) for n in (1, 2, 3, 4)}
for i in range(int(os.environ.get('CHURN_CALLS', 300))):
    n = 1 + i % 4
    out = G.synthesis(wss[n], c=c.repeat(n, 1), ray_jitter=False, return_seg=True)
    if i % 20 == 0:
        torch.cuda.synchronize()
        print(i, dict(graph_cache.STATS), torch.cuda.memory_allocated() >> 20, 'MiB', flush=True)
torch.cuda.synchronize()
print('OK', mode, dict(graph_cache.STATS))
