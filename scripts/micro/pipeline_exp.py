"""EXPERIMENT (round 3): a two-stage software pipeline over batches inside one captured graph.
Replay k runs, side by side on two streams, mapping + backbone of batch k (latency-bound low-resolution layers that leave most CUs
idle) and ray-marcher + super-resolution of batch k - 1 (from the tri-planes replay k - 1 left in a static buffer).
Prints ms per replay of the plain GraphedRenderer and of the pipelined one, and checks that replay k + 1 returns the frames of batch k."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd')); sys.path.insert(0, ROOT)
import torch
from training import triplane, networks
from torch_utils import hip_plugin

dev = torch.device('cuda:0')
hip_plugin.load()
hip_plugin.conv_arithmetic(os.environ.get('ARITH', 'f16x3'))
torch.manual_seed(0)
G = triplane.TriPlaneGenerator(None).eval().to(dev)
B = 4


class Pipelined(triplane.GraphedRenderer):
    def __init__(self, G, batch, device, **kw):
        sp = G.synthesis.spec
        self._sa = torch.cuda.Stream(device=device)
        self.ws_cur = torch.zeros([batch, G.num_ws, G.w_dim], device=device)
        pc = 3 * sp.plane_channels
        mk = lambda: torch.zeros([batch, pc, sp.plane_resolution, sp.plane_resolution], device=device).contiguous(memory_format=torch.channels_last)
        self.planes_cur = (mk(), mk())
        self.cam_cur = triplane.conditioning_label(device).repeat(batch, 1)
        self.jit_cur = torch.rand([batch, sp.render_size ** 2, sp.num_steps], device=device)
        super().__init__(G, batch, device, **kw)

    def _body(self):
        G = self.G
        main = torch.cuda.current_stream()
        a = self._sa
        a.wait_stream(main)
        with torch.cuda.stream(a):
            ws_new = G.mapping(self.z, self.c_cond, truncation_psi=self.psi)
            voxel_ws, _ = G.synthesis.split_ws(ws_new)
            with networks.amax_arena(ws_new.shape[0], ws_new.device):
                img_v, seg_v = G.synthesis.backbone(voxel_ws, noise_mode=self.noise_mode)
        out = G.synthesis(self.ws_cur, c=self.cam_cur, noise_mode=self.noise_mode, return_seg=True, ray_jitter=self.jit_cur,
                          cached_planes=self.planes_cur)
        main.wait_stream(a)
        for t in (ws_new, img_v, seg_v):
            t.record_stream(main)
        self.planes_cur[0].copy_(img_v); self.planes_cur[1].copy_(seg_v)
        self.ws_cur.copy_(ws_new); self.cam_cur.copy_(self.c_cam); self.jit_cur.copy_(self.jitter)
        return out


def timeit(r, n=60):
    z = torch.randn(B, G.z_dim, device=dev)
    cond = triplane.conditioning_label(dev).repeat(B, 1)
    cams = torch.cat([triplane.camera_label(y, device=dev) for y in (-0.5, -0.15, 0.2, 0.5)])
    for _ in range(10):
        r(z, cond, cams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r(z, cond, cams)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


plain = triplane.GraphedRenderer(G, B, dev)
pipe = Pipelined(G, B, dev)
for rep in range(2):
    a, b = timeit(plain), timeit(pipe)
    print(f'plain {a:.3f} ms / replay ({B / a * 1e3:.0f} frames/s)   pipelined {b:.3f} ms / replay ({B / b * 1e3:.0f} frames/s)')
# semantics: replay k + 1 of the pipeline returns the frames of the inputs of replay k
cond = triplane.conditioning_label(dev).repeat(B, 1)
cams = torch.cat([triplane.camera_label(y, device=dev) for y in (-0.5, -0.15, 0.2, 0.5)])
z1 = torch.randn(B, G.z_dim, device=dev); z2 = torch.randn(B, G.z_dim, device=dev)
u = torch.rand_like(plain.jitter)
ref = [t.clone() for t in plain(z1, cond, cams, jitter=u)]
pipe(z1, cond, cams, jitter=u)
got = [t.clone() for t in pipe(z2, cond, cams, jitter=u)]
print('pipelined output of batch k at replay k + 1 equals the plain renderer:', all(torch.equal(x, y) for x, y in zip(ref, got)),
      'max abs diff', max(float((x - y).abs().max()) for x, y in zip(ref, got)))
