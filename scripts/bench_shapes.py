"""BASELINE config 5: extract_shapes.py density cube (256^3 lattice, 0.9 scale, 1 seed) on one GPU.
Prints one JSON line: Mvoxel/s of the query loop (tri-planes resident), and of the whole driver (mapping + backbone + loop)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd')); sys.path.insert(0, ROOT)
import numpy as np
import torch
from training import shape_extraction as se, triplane
from torch_utils import hip_plugin

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda:0')
torch.manual_seed(0)
G = triplane.TriPlaneGenerator().eval().to(dev)
z = torch.from_numpy(np.random.RandomState(0).randn(1, 512)).float().to(dev)
c = triplane.conditioning_label(dev)
with torch.no_grad():
    ws = G.mapping(z, c, truncation_psi=0.5)
    img_v, seg_v = se.triplanes_from_ws(G.synthesis, ws, noise_mode='const')
res = {}
for name, mb, mat in (('reference_style_materialized_chunks_1e5', 100000, True), ('chunked_1e5', 100000, False), ('single_launch', None, False)):
    for _ in range(2):
        se.density_cube(G.synthesis.renderer, img_v, seg_v, N, max_batch=mb, materialize=mat)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        cube = se.density_cube(G.synthesis.renderer, img_v, seg_v, N, max_batch=mb, materialize=mat)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    res[name] = dict(ms=dt * 1e3, mvoxel_per_s=N ** 3 / dt / 1e6)
torch.cuda.synchronize(); t0 = time.perf_counter()
cube = se.sample_generator_ide3d(G, None, z, c, voxel_resolution=N, to_numpy=False, noise_mode='const')
torch.cuda.synchronize(); dt = time.perf_counter() - t0
res['driver_incl_mapping_backbone'] = dict(ms=dt * 1e3, mvoxel_per_s=N ** 3 / dt / 1e6)
# kernel-only: one density launch with the lattice generated in registers (HIP events on the launch stream)
R = G.synthesis.renderer
corner, vs = np.array([0., 0., 0.]) - 1.0, 2.0 / (N - 1)
R.density_lattice(img_v, seg_v, N, vs, corner, 0.9, 0, N ** 3)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); R.density_lattice(img_v, seg_v, N, vs, corner, 0.9, 0, N ** 3); b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b)
bytes_algo = N ** 3 * 4 + 96 * 256 * 256 * 4          # sigma out + the semantic tri-plane (coordinates are generated)
res['kernel_density_lattice'] = dict(ms=ms, mvoxel_per_s=N ** 3 / ms / 1e3, algorithmic_GBps=bytes_algo / ms / 1e6,
                                     mlp_tflops=N ** 3 * 2 * (32 * 64 + 64 * 20) / ms / 1e9)
print(json.dumps(dict(metric='extract_shapes density cube', voxels=N ** 3, finite=bool(torch.isfinite(cube).all()), **res,
                      native_launches=dict(hip_plugin.CALLS))))
