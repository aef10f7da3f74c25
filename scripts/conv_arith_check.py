"""Accuracy and speed of the three arithmetics of the shared-weight 3x3 layers (fp32 MFMA / bf16x6 / bf16x3).

    python scripts/conv_arith_check.py [--time] [--json out.json]

Accuracy: every arithmetic against a float64 convolution of the same operands (ATen, on the GPU) on ragged shapes;
reported as max |err| / max |ref| and rms err / rms ref.  --time: the eight 3x3 layers of the kernel table per arithmetic.
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT, os.path.join(ROOT, 'scripts')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

ARITH = {'fp32': 1, 'bf16x6': 6, 'bf16x3': 3}


def ref64(x, w, s, d, mode):
    xs = (x * s[:, :, None, None]).double()          # the kernels round x * s to fp32 once, like this
    if mode == 0:
        y = F.conv2d(xs, w.double(), padding=1)
    else:
        y = F.conv_transpose2d(xs, w.double().transpose(0, 1), stride=2)
    return y * d.double()[:, :, None, None]


def accuracy(device):
    from torch_utils import hip_plugin
    mc = hip_plugin.ModconvPlugin.modconv2d
    g = torch.Generator().manual_seed(7)
    rn = lambda *sh: torch.randn(*sh, generator=g).to(device)
    rows = []
    for (n, cin, cout, h, w, mode) in ((2, 40, 72, 37, 45, 0), (1, 128, 128, 64, 64, 0), (3, 64, 200, 33, 20, 0), (2, 512, 64, 16, 16, 0), (4, 128, 128, 256, 256, 0), (4, 64, 64, 512, 512, 0),
                                       (2, 40, 72, 37, 45, 2), (1, 128, 64, 64, 64, 2), (2, 96, 130, 20, 33, 2), (4, 512, 512, 16, 16, 2), (4, 128, 64, 256, 256, 2), (4, 256, 128, 128, 128, 2)):
        x = rn(n, cin, h, w); wt = rn(cout, cin, 3, 3) / math.sqrt(cin * 9); s = rn(n, cin) + 1; d = torch.rand(n, cout, generator=g).to(device) + 0.5
        ref = ref64(x, wt, s, d, mode)
        row = dict(shape=[n, cin, cout, h, w], mode=mode)
        for name, a in ARITH.items():
            y = mc(x, wt, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=mode, arith=a).double()
            err = (y - ref)
            row[name] = dict(max_rel=float(err.abs().max() / ref.abs().max()), rms_rel=float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
        rows.append(row)
        print(row['shape'], 'mode', mode, ' '.join(f"{k}: max {v['max_rel']:.2e} rms {v['rms_rel']:.2e}" for k, v in row.items() if isinstance(v, dict)), flush=True)
    return rows


def timing(device, iters):
    from torch_utils import hip_plugin
    import kernel_rooflines as kr
    mc = hip_plugin.ModconvPlugin.modconv2d
    g = torch.Generator().manual_seed(3)
    rn = lambda *sh: torch.randn(*sh, generator=g).to(device)
    rows = []
    for tag, cin, cout, res, mode in (('3x3 128->128 @256', 128, 128, 256, 0), ('3x3 256->256 @128', 256, 256, 128, 0), ('3x3 512->512 @64', 512, 512, 64, 0),
                                      ('3x3 64->64 @512', 64, 64, 512, 0), ('3x3 512->512 @32', 512, 512, 32, 0), ('3x3 512->512 @16', 512, 512, 16, 0),
                                      ('transposed 512->256 in@64', 512, 256, 64, 2), ('transposed 256->128 in@128', 256, 128, 128, 2),
                                      ('transposed 128->64 in@256', 128, 64, 256, 2), ('transposed 512->512 in@32', 512, 512, 32, 2),
                                      ('transposed 512->512 in@16', 512, 512, 16, 2), ('transposed 32->128 in@128', 32, 128, 128, 2)):
        xx = rn(kr.N, cin, res, res); ww = rn(cout, cin, 3, 3); ss = rn(kr.N, cin) + 1; dc = torch.rand(kr.N, cout, generator=g).to(device)
        nzz = rn(res, res); bz = rn(cout)
        flops = 2 * cin * cout * 9 * res * res * kr.N
        row = dict(case=tag)
        for name, a in ARITH.items():
            if mode == 0:
                fn = lambda a=a: mc(xx, ww, ss, dc, nzz, 1.0, bz, 3, 0.2, math.sqrt(2), -1.0, arith=a)
            else:
                fn = lambda a=a: mc(xx, ww, ss, dc, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=2, arith=a)
            us = kr._time(fn, iters, device)
            row[name] = dict(us=us, tflops=flops / us / 1e6)
        rows.append(row)
        print(f"{tag:32s} " + '  '.join(f"{k}: {v['us']:7.1f} us {v['tflops']:6.1f} TF" for k, v in row.items() if isinstance(v, dict)), flush=True)
    return rows


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--time', action='store_true'); ap.add_argument('--iters', type=int, default=10); ap.add_argument('--json')
    ap.add_argument('--no-accuracy', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    out = {}
    if not a.no_accuracy:
        out['accuracy'] = accuracy(dev)
    if a.time:
        out['timing'] = timing(dev, a.iters)
    if a.json:
        json.dump(out, open(a.json, 'w'), indent=1)
