"""Time csrc/upfirdn2d.hip alone on the generator's two FIR shapes (after the transposed conv: 4x4, up 1, pad 1, gain 4;
skip-image upsample: 4x4, up 2) — target for rocprofv3.  Prints achieved GB/s against (in + out) * 4 bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd')); sys.path.insert(0, ROOT)
import torch
from torch_utils.ops import upfirdn2d
dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
for name, shape, kw in (('post-tconv 64ch 513->512', (4, 64, 513, 513), dict(padding=[1, 1, 1, 1], gain=4)),
                        ('post-tconv 128ch 257->256', (4, 128, 257, 257), dict(padding=[1, 1, 1, 1], gain=4)),
                        ('post-tconv 512ch 65->64', (4, 512, 65, 65), dict(padding=[1, 1, 1, 1], gain=4)),
                        ('skip up2 96ch 128->256', (4, 96, 128, 128), dict(up=2, padding=[2, 1, 2, 1], gain=4)),
                        ('skip up2 22ch 256->512', (4, 22, 256, 256), dict(up=2, padding=[2, 1, 2, 1], gain=4))):
    x = torch.randn(*shape, device=dev)
    y = upfirdn2d.upfirdn2d(x, f, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y = upfirdn2d.upfirdn2d(x, f, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = (x.numel() + y.numel()) * 4
    print(f'{name}: {ms * 1e3:.1f} us, {nbytes / ms / 1e6:.0f} GB/s ({nbytes / ms / 1e6 / 80:.1f}% of 8 TB/s)')
