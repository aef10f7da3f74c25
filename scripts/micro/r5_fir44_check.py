"""fir44_kernel (lean fp32 4x4 FIR at unit rate) against the tile kernel it replaces: run once with IDE3D_FIR_NO_LEAN=1 (tile kernel) writing
its outputs to a file, then normally with that file as the reference: every tensor must be bit-equal.  Usage: r5_fir44_check.py <out.pt> [ref.pt]"""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'ide-3d_amd'))
import torch
from torch_utils import hip_plugin
from torch_utils.ops import upfirdn2d

dev = torch.device('cuda:0')
upfirdn2d._init()
g = torch.Generator().manual_seed(3)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
f4 = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
outs = {}
for name, (n, c, h, w) in {'513': (2, 16, 513, 513), '257': (2, 8, 257, 257), 'ragged': (1, 5, 77, 201), '129': (3, 7, 129, 129)}.items():
    x = torch.nn.functional.pad(rn(n, c, h, w), (0, (-w) % 4))[..., :w]
    oh, ow = h - 1, w - 1
    nz = rn(oh, ow); bb = rn(c)
    am = torch.zeros(n, hip_plugin.AMAX_FLOATS, device=dev)
    for tag, kw in {'plain': {}, 'epi': dict(noise=nz, noise_strength=0.7, bias=bb, act=3, alpha=0.2, act_gain=math.sqrt(2), clamp=-1.0, y_amax=am),
                    'clamp': dict(noise=nz, noise_strength=0.7, bias=bb, act=3, alpha=0.2, act_gain=math.sqrt(2), clamp=0.8),
                    'lin': dict(bias=bb, act=1, alpha=0.0, act_gain=1.0, clamp=-1.0), 'alpha2': dict(bias=bb, act=3, alpha=1.5, act_gain=1.0, clamp=-1.0)}.items():
        for flip in (False, True):
            y = upfirdn2d._plugin.upfirdn2d_ex(x, f4, 1, 1, 1, 1, 1, 1, 1, 1, flip, 4.0, **kw)
            outs[f'{name}/{tag}/{int(flip)}'] = y.cpu()
    outs[f'{name}/amax'] = am.amax(dim=1).cpu()
    xr = x.double(); ff = f4.double() * 4.0
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xr, (1, 1, 1, 1)).reshape(-1, 1, h + 2, w + 2), ff.flip(0, 1)[None, None]).reshape(n, c, oh, ow)
    refe = torch.nn.functional.leaky_relu(ref + nz.double() * 0.7 + bb.double()[None, :, None, None], 0.2) * math.sqrt(2)
    e0 = float((outs[f'{name}/plain/0'].double() - ref.cpu()).abs().max() / ref.abs().max())
    e1 = float((outs[f'{name}/epi/0'].double() - refe.cpu()).abs().max() / refe.abs().max())
    print(name, 'err vs float64: plain %.2e  epilogue %.2e' % (e0, e1), ' amax ok', torch.equal(outs[f'{name}/amax'], outs[f'{name}/epi/1'].abs().amax(dim=(1, 2, 3))))
# the skip upsampler (up 2, pad (2, 1, 2, 1), gain 4), plain and with the skip operand
for name, (n, c, h, w) in {'u128': (2, 12, 128, 128), 'u256': (1, 22, 256, 256), 'u8': (3, 5, 8, 8), 'uragged': (2, 3, 20, 36), 'u64x200': (1, 4, 64, 200)}.items():
    x = rn(n, c, h, w); add = rn(n, c, 2 * h, 2 * w)
    for tag, kw in {'plain': {}, 'add': dict(add=add), 'addact': dict(add=add, bias=rn(c), act=3, alpha=0.2, act_gain=1.3, clamp=0.9)}.items():
        for flip in (False, True):
            y = upfirdn2d._plugin.upfirdn2d_ex(x, f4, 2, 2, 1, 1, 2, 1, 2, 1, flip, 4.0, **kw)
            outs[f'{name}/{tag}/{int(flip)}'] = y.cpu()
    up = torch.zeros(n, c, 2 * h, 2 * w, dtype=torch.float64, device=dev); up[:, :, ::2, ::2] = x.double()
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(up, (2, 1, 2, 1)).reshape(-1, 1, 2 * h + 3, 2 * w + 3), (f4.double() * 4.0).flip(0, 1)[None, None]).reshape(n, c, 2 * h, 2 * w)
    e = float((outs[f'{name}/add/0'].double() - (ref + add.double()).cpu()).abs().max() / ref.abs().max())
    print(name, 'up2 + add err vs float64: %.2e' % e)
torch.save(outs, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    bad = [k for k in outs if not torch.equal(outs[k], ref[k])]
    print('compared', len(outs), 'tensors bit for bit; different:', bad)
print('lean kernel active:', os.environ.get('IDE3D_FIR_NO_LEAN') is None)
