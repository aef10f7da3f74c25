"""Developer aid: which framework (non-ide3d) kernels still run in one eager step, and from which Python lines."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd')); sys.path.insert(0, ROOT)
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
from training import triplane, distributed_render as dr

dev = torch.device('cuda:0')
torch.manual_seed(0)
G = triplane.TriPlaneGenerator().eval().to(dev)
cond = triplane.conditioning_label(dev).repeat(4, 1)
cams = torch.cat([triplane.camera_label(y, device=dev) for y in (-0.5, 0.0, 0.5, 0.25)])
pal = dr.palette_tensor(19, dev)
z = torch.randn(4, 512, device=dev)


def step():
    with torch.no_grad():
        ws = G.mapping(z, cond)
        img, seg = G.synthesis(ws, c=cams, noise_mode='const', return_seg=True)
        return dr.frames_u8(img, seg, pal)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = collections.Counter(); where = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.device_type.name == 'CPU' and ev.name.startswith('aten::') and ev.name not in ('aten::empty', 'aten::view', 'aten::reshape', 'aten::as_strided',
            'aten::select', 'aten::slice', 'aten::narrow', 'aten::expand', 'aten::unsqueeze', 'aten::squeeze', 'aten::t', 'aten::transpose', 'aten::permute',
            'aten::empty_like', 'aten::empty_strided', 'aten::to', 'aten::_to_copy', 'aten::detach', 'aten::alias', 'aten::result_type', 'aten::item', 'aten::stride',
            'aten::contiguous', 'aten::flatten', 'aten::unflatten', 'aten::_unsafe_view', 'aten::view_as', 'aten::size', 'aten::is_nonzero', 'aten::lift_fresh',
            'aten::resolve_conj', 'aten::resolve_neg', 'aten::numpy_T', 'aten::chunk', 'aten::split', 'aten::unbind', 'aten::index_select', 'aten::zeros', 'aten::ones'):
        rows[ev.name] += 1
        st = [s for s in (ev.stack or []) if 'ide-3d_amd' in s]
        where[ev.name][st[0].split('ide-3d_amd/')[-1] if st else '?'] += 1
for name, cnt in rows.most_common(25):
    print(f'{cnt:4d} {name}')
    for w, c in where[name].most_common(6):
        print(f'        {c:3d}  {w}')
