"""Row-parity pair form of the transposed 3x3 convolution (modconv_split_pair_kernel) against the all-class form: bit-equality on ragged
shapes (IDE3D_MODCONV_PAIR=2 forces the form wherever it exists, =0 switches it off) and time per launch on the model's layers."""
import os, sys, math, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'ide-3d_amd'))
import torch
from torch_utils import hip_plugin

dev = torch.device('cuda:0')
ARITH = {'bf16x6': 6, 'bf16x3': 3, 'f16x3': 16}


def amax_slots(x):
    out = torch.zeros(x.shape[0], hip_plugin.AMAX_FLOATS, device=x.device)
    out[:, 5 * 64] = x.abs().amax(dim=(1, 2, 3))
    return out


def run(x, w, s, d, arith, knob, pad_rows=True, xam=None, am=None):
    if knob is None: os.environ.pop('IDE3D_MODCONV_PAIR', None)
    else: os.environ['IDE3D_MODCONV_PAIR'] = str(knob)
    if am is None: am = torch.zeros(x.shape[0], hip_plugin.AMAX_FLOATS, device=dev)
    y = hip_plugin.ModconvPlugin.modconv2d(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=2, arith=arith, x_amax=amax_slots(x) if xam is None else xam, y_amax=am, pad_rows=pad_rows)
    return y, am


def plan(shape, arith, knob):
    if knob is None: os.environ.pop('IDE3D_MODCONV_PAIR', None)
    else: os.environ['IDE3D_MODCONV_PAIR'] = str(knob)
    n, cin, cout, h, w = shape
    pl = hip_plugin.modconv_plan(n, cin, cout, h, w, k=3, mode=2, arith=arith, epilogue='plain', x_amax=True)
    return {k: pl[k] for k in ('kind', 'tile_h', 'tile_w', 'rows', 'waves', 'workgroups', 'split_k', 'strip')}


bad = 0
for shape in [(2, 64, 128, 20, 24), (1, 48, 64, 33, 17), (3, 80, 130, 16, 16), (2, 128, 64, 64, 40), (1, 256, 128, 37, 50), (2, 32, 128, 31, 33), (4, 256, 128, 128, 128), (4, 128, 64, 256, 256)]:
    n, cin, cout, h, w_ = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w_, generator=g).to(dev); wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).to(dev)
    s = (torch.randn(n, cin, generator=g) + 1).to(dev); d = (torch.rand(n, cout, generator=g) + 0.5).to(dev)
    ref = None
    if n * cin * cout * h * w_ < 3e8:
        ref = torch.nn.functional.conv_transpose2d((x.double() * s.double()[:, :, None, None]), wt.double().transpose(0, 1), stride=2) * d.double()[:, :, None, None]
    for name, code in ARITH.items():
        y0, a0 = run(x, wt, s, d, code, 0)
        y2, a2 = run(x, wt, s, d, code, 2)
        eq = torch.equal(y0, y2) and torch.equal(a0.amax(dim=1), a2.amax(dim=1))
        err = float((y2.double() - ref).abs().max() / ref.abs().max()) if ref is not None else float('nan')
        bad += (not eq)
        print(f'{shape} {name}: pair == all-class {eq}  maxdiff {float((y0 - y2).abs().max()):.3e}  err vs f64 {err:.2e}  plan0 {plan(shape, code, 0)}  plan2 {plan(shape, code, 2)}', flush=True)
print('MISMATCHES', bad)


def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for shape in [(4, 256, 128, 128, 128), (4, 128, 64, 256, 256), (4, 512, 256, 64, 64), (4, 32, 128, 128, 128), (4, 512, 512, 32, 32), (1, 128, 64, 256, 256), (1, 256, 128, 128, 128)]:
    n, cin, cout, h, w_ = shape
    x = torch.randn(n, cin, h, w_, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(9 * cin)
    s = torch.randn(n, cin, device=dev) + 1; d = torch.rand(n, cout, device=dev) + 0.5
    xam = amax_slots(x); am = torch.zeros(n, hip_plugin.AMAX_FLOATS, device=dev)
    for name, code in (('bf16x6', 6), ('f16x3', 16)):
        row = []
        for knob in (0, None, 2, '2L'):
            if knob == '2L':
                os.environ['IDE3D_MODCONV_NO_R16'] = '1'; knob = None
            t = timeit(lambda: run(x, wt, s, d, code, knob, xam=xam, am=am))
            os.environ.pop('IDE3D_MODCONV_NO_R16', None)
            row.append(f'PAIR={knob}: {t:7.1f} us ({plan(shape, code, knob)["workgroups"]} wgs, tile_h {plan(shape, code, knob)["tile_h"]})')
        print(shape, name, ' | '.join(row), flush=True)
