"""Developer aid: phase timing of the ray-grid gather kernel.  Needs a library built with -DIDE3D_TT_TRACE:
    make -C ide-3d_amd/csrc clean; make -C ide-3d_amd/csrc -j8 EXTRA=-DIDE3D_TT_TRACE   (see triplane_tile.hip)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd')); sys.path.insert(0, ROOT)
import torch
import bench
from torch_utils import hip_plugin
r = bench.bench_gather(torch.device('cuda:0'), iters=3)
print('avg launch us', r['avg_launch_us'])
buf = (ctypes.c_ulonglong * 256)()
assert hip_plugin.load().ide3d_debug_tt(buf) == 0
v = list(buf)
print('chunk: cycles of [bbox reduce, barrier 1, region table, tap table + region fetch + LDS fill, barrier 2, blend]')
for ch in range(24):
    row = v[ch * 8:ch * 8 + 7]
    if row[0]:
        print(ch, [row[i + 1] - row[i] for i in range(6)], 'total', row[6] - row[0])
