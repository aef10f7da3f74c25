import sys, time, math, os
sys.path.insert(0, '/root/repo/ide-3d_amd'); sys.path.insert(0, '/root/repo')
import torch
from training.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
dev = torch.device('cuda:0')
def pose(yaw):
    camera_points, phi, theta = sample_camera_positions(dev, n=1, r=2.7, horizontal_mean=yaw + math.pi * 0.5, vertical_mean=math.pi * 0.5, mode=None)
    c = create_cam2world_matrix(-camera_points, camera_points, device=dev)
    c = c.reshape(1, -1)
    c = torch.cat((c, torch.tensor([4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).reshape(1, -1).to(c)), -1)
    return c
for _ in range(20): pose(0.1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(300): pose(0.001 * i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'pose: host {1e6 * (t1 - t0) / 300:.1f} us per pose, incl. final sync {1e6 * (t2 - t0) / 300:.1f}')
