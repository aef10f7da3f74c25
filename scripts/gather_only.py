"""Run only the stand-alone tri-plane gather (benchmark shape) a few times — target for rocprofv3 PMC passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd')); sys.path.insert(0, ROOT)
import torch
import bench
r = bench.bench_gather(torch.device('cuda:0'), iters=int(sys.argv[1]) if len(sys.argv) > 1 else 10,
                       tiled=not (len(sys.argv) > 2 and sys.argv[2] == 'flat'),
                       warm_launches=int(sys.argv[3]) if len(sys.argv) > 3 else 400)
print(r)
