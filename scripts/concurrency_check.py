"""Bit-stability of every kernel of the library while a matrix-core convolution runs on another stream.

Background (measured on MI355X, round 2): a wave executing packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 /
v_pk_add_f32) can return wrong results while ANOTHER kernel's wave on the same SIMD executes v_mfma_f32_32x32x16_bf16.  The
library is therefore built without packed fp32 instructions (csrc/Makefile); this script is the check.

    python scripts/concurrency_check.py [--arith bf16x6,bf16x3,fp32] [--only substr] [--reps 40]

For every case of scripts/kernel_rooflines.py: the result computed alone is the reference; then the case is launched `reps`
times on stream B while stream A loops a convolution, and every result is compared bitwise.
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT, os.path.join(ROOT, 'scripts')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402


def _flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in _flat(x)]
    if isinstance(o, dict):
        return [t for x in o.values() for t in _flat(x)]
    return []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arith', default='bf16x6,bf16x3,fp32'); ap.add_argument('--only', default=''); ap.add_argument('--reps', type=int, default=40)
    a = ap.parse_args()
    import kernel_rooflines as kr
    from torch_utils import hip_plugin
    dev = torch.device('cuda', 0)
    mc = hip_plugin.ModconvPlugin.modconv2d
    g = torch.Generator().manual_seed(5)
    rn = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    x = rn(4, 64, 512, 512); wt = rn(64, 64, 3, 3); s = rn(4, 64) + 1; d = torch.rand(4, 64, generator=g).to(dev)
    xh = rn(4, 64, 512, 512); wh = rn(4, 22, 64, 1, 1); bh = rn(22)
    codes = {'fp32': 1, 'bf16x3': 3, 'bf16x6': 6}
    aggressors = [(f'3x3 64->64 @512 {n}', (lambda c=codes[n]: mc(x, wt, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, arith=c))) for n in a.arith.split(',')]
    aggressors.append(('1x1 heads 64->22 @512 fp32 MFMA', lambda: mc(xh, wh, None, None, None, 0.0, bh, 1, 0.0, 1.0, 256.0)))
    cases = [c for c in kr.cases(dev) if a.only in c[0] and 'density_lattice' not in c[0]]
    sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    total_bad = 0
    for aname, afn in aggressors:
        with torch.cuda.stream(sA):
            afn()
        torch.cuda.synchronize()
        for name, _frag, _bound, _amount, fn in cases:
            with torch.cuda.stream(sB), torch.no_grad():
                ref = [t.clone() for t in _flat(fn())]
            torch.cuda.synchronize()
            outs = []
            with torch.cuda.stream(sA):
                for _ in range(max(4, a.reps // 2)):
                    afn()
            with torch.cuda.stream(sB), torch.no_grad():
                for _ in range(a.reps):
                    outs.append([t for t in _flat(fn())])
            torch.cuda.synchronize()
            bad = sum(1 for o in outs if not all(torch.equal(u, v) for u, v in zip(o, ref)))
            total_bad += bad
            print(f'{"CORRUPT" if bad else "ok     "} {bad:3d}/{a.reps}  victim: {name[:70]:70s} beside: {aname}', flush=True)
            del outs
    print('total corrupted launches:', total_bad)
    return 1 if total_bad else 0


if __name__ == '__main__':
    sys.exit(main())
