"""BASELINE config 4: 64 seeds x 8 camera poses, one rank per GPU, uint8 frames gathered on rank 0.
Run on one GPU as `python scripts/bench_config4.py`, or on N GPUs with
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_config4.py`.
Prints one JSON line (rank 0): images/s of the whole job, with the pose-independent tri-planes cached per seed."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd')); sys.path.insert(0, ROOT)
import numpy as np
import torch
from training import triplane, distributed_render as dr

world, rank, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
dev = torch.device('cuda', local); torch.cuda.set_device(dev)
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('nccl', device_id=dev)
torch.manual_seed(0)
G = triplane.TriPlaneGenerator().eval().to(dev)
seeds, yaws = list(range(64)), list(np.linspace(-0.5, 0.5, 8))
res = {}
for name, cache in (('cached_triplanes', True), ('full_synthesis', False)):
    dr.render_grid_sharded(G, seeds[:4 * world], yaws, dev, rank, world, cache_backbone=cache)      # warm-up: weight packing, plugin init, and the second sighting of the call signatures (G.synthesis captures its hipGraph there)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    frames = dr.render_grid_sharded(G, seeds, yaws, dev, rank, world, cache_backbone=cache)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[name] = dict(seconds=dt, images_per_s=len(seeds) * len(yaws) / dt)
if rank == 0:
    print(json.dumps(dict(metric='64 seeds x 8 poses, 512x512 RGB+seg uint8 frames gathered on rank 0', n_gpus=world,
                          frames_shape=list(frames.shape), **res)))
if world > 1:
    dist.destroy_process_group()
