"""Writes scripts/micro/bin/gather_coords.bin: float32 [4, 393216, 3], the sample positions of bench.py's `bench_gather`
(64 x 64 rays x 96 jittered depth steps of four cameras), computed on the CPU with the same host code.  The file is a build
product (git-ignored) that travels to the GPU box with the snapshot; scripts/micro/gather_bench.cpp reads it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd')); sys.path.insert(0, ROOT)
import torch
from training import triplane, volumetric_rendering as vr
g = torch.Generator().manual_seed(0)
n = 4
torch.randn(n, 96, 256, 256, generator=g)          # bench_gather draws the planes first: keep the jitter stream identical
pts, z, d = vr.get_initial_rays_trig(n, 96, 'cpu', 18.0, (64, 64), 2.25, 3.3)
yaws = tuple(float(v) for v in os.environ.get('GATHER_YAWS', '-0.5,-0.15,0.2,0.5').split(','))          # bench_gather's cameras by default
cam = torch.cat([triplane.camera_label(y) for y in yaws])[:, :16].reshape(-1, 4, 4)
wp, *_ = vr.transform_sampled_points(pts, z, d, 'cpu', h_stddev=0, v_stddev=0, camera=cam, mode=None, jitter=torch.rand(z.shape, generator=g))
out = os.path.join(ROOT, 'scripts', 'micro', 'bin')
os.makedirs(out, exist_ok=True)
name = os.environ.get('GATHER_COORDS_NAME', 'gather_coords.bin')
wp.reshape(n, -1, 3).contiguous().numpy().tofile(os.path.join(out, name))
print('wrote', os.path.join(out, name))
