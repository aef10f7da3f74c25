"""Is the f16x3 convolution slower than bf16x3 because of its instructions or because of its DATA (11-bit vs 8-bit significands ->
more multiplier activity -> lower clocks under the power cap)?  Times the same launches on (a) N(0,1) operands, (b) operands rounded
to 8 significant bits (every low piece is zero), (c) zeros.   python scripts/micro/f16_power_check.py"""
import os, sys, math, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ide-3d_amd'))
import torch
from torch_utils import hip_plugin
dev = torch.device('cuda')
mc = hip_plugin.ModconvPlugin.modconv2d
g = torch.Generator().manual_seed(0)

def timeit(fn, iters=30):
    for _ in range(5): fn()
    st = torch.cuda.Stream(dev); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn(); graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            for _ in range(iters): fn()
    torch.cuda.synchronize()
    for _ in range(3): graph.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 / iters * 1e6

def round8(t):
    return (t.view(torch.int32) & ~0xFFFF).view(torch.float32)      # keep sign, exponent, 7 mantissa bits (= one bf16 piece)

for tag, cin, cout, res, mode in (('3x3 128->128 @256', 128, 128, 256, 0), ('3x3 256->256 @128', 256, 256, 128, 0), ('tconv 256->128 in@128', 256, 128, 128, 2), ('tconv 128->64 in@256', 128, 64, 256, 2)):
    n = 4
    for data in ('normal', 'round8', 'zeros'):
        x = torch.randn(n, cin, res, res, generator=g).to(dev); w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(dev)
        s = torch.ones(n, cin, device=dev); d = torch.ones(n, cout, device=dev)
        if data == 'round8': x, w = round8(x), round8(w)
        if data == 'zeros': x, w = torch.zeros_like(x), torch.zeros_like(w)
        xam = x.abs().amax(dim=(1, 2, 3))[:, None].repeat(1, hip_plugin.AMAX_FLOATS).contiguous()
        row = []
        for name, code in (('fp32', 1), ('bf16x6', 6), ('bf16x3', 3), ('f16x3', 16)):
            t = timeit(lambda: mc(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=mode, arith=code, x_amax=xam))
            row.append(f'{name} {t:7.1f}')
        print(f'{tag:24s} {data:7s} | ' + ' | '.join(row), flush=True)
