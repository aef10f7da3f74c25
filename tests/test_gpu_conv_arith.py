"""The three arithmetics of the shared-weight 3x3 layers (fp32 MFMA / bf16x6 / bf16x3, include/ide3d_hip.h) and their
coexistence with the rest of the library on one GPU.

 * accuracy: each arithmetic against a float64 convolution (ATen on the GPU) on ragged and full-size shapes.  Stated tolerances
   (relative to max |ref|): fp32 and bf16x6 4e-6 (both are fp32-grade: exact / <= 2^-23 products, fp32 accumulation),
   bf16x3 3e-5 (products to ~2^-17);
 * bf16x6 must be no worse than 3x the fp32 MFMA's own rounding error, rms;
 * launches are bit-reproducible, on any stream and inside a hipGraph;
 * no kernel of the library changes its result while a bf16 matrix-core convolution runs on another stream (the packed-fp32 /
   v_mfma_f32_32x32x16_bf16 interaction found in round 2: csrc/Makefile, scripts/concurrency_check.py).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ARITH = {'fp32': 1, 'bf16x6': 6, 'bf16x3': 3}
TOL = {'fp32': 4e-6, 'bf16x6': 4e-6, 'bf16x3': 3e-5}
SHAPES = [  # n, cin, cout, h, w, mode
    (2, 40, 72, 37, 45, 0), (3, 64, 200, 33, 20, 0), (2, 512, 64, 16, 16, 0), (4, 128, 128, 256, 256, 0), (1, 17, 130, 12, 300, 0),
    (2, 40, 72, 37, 45, 2), (2, 96, 130, 20, 33, 2), (4, 512, 512, 16, 16, 2), (4, 128, 64, 256, 256, 2), (1, 33, 64, 13, 12, 2),
]


def _mc():
    from torch_utils import hip_plugin
    return hip_plugin.ModconvPlugin.modconv2d


def _operands(shape, dev, seed=7):
    n, cin, cout, h, w, mode = shape
    g = torch.Generator().manual_seed(seed)
    rn = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    return rn(n, cin, h, w), rn(cout, cin, 3, 3) / math.sqrt(cin * 9), rn(n, cin) + 1, torch.rand(n, cout, generator=g).to(dev) + 0.5


def _ref64(x, w, s, d, mode):
    xs = (x * s[:, :, None, None]).double()          # the kernels round x * s to fp32 once
    y = F.conv2d(xs, w.double(), padding=1) if mode == 0 else F.conv_transpose2d(xs, w.double().transpose(0, 1), stride=2)
    return y * d.double()[:, :, None, None]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_arithmetics_vs_float64(gpu_device, shape):
    x, w, s, d = _operands(shape, gpu_device)
    ref = _ref64(x, w, s, d, shape[5])
    rms = {}
    for name, code in ARITH.items():
        y = _mc()(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=shape[5], arith=code).double()
        err = y - ref
        assert float(err.abs().max() / ref.abs().max()) < TOL[name], f'{name}: {float(err.abs().max() / ref.abs().max()):.3e}'
        rms[name] = float(err.pow(2).mean().sqrt())
    assert rms['bf16x6'] < 3 * rms['fp32'], f'bf16x6 is not fp32-grade here: rms {rms}'


def test_epilogue_and_split_k_agree_between_arithmetics(gpu_device):
    """noise + bias + lrelu + gain + clamp on the split-bf16 loop (and its split-K reduction at 16x16) against the fp32 loop."""
    for shape in ((4, 512, 512, 16, 16, 0), (2, 64, 64, 40, 52, 0)):
        x, w, s, d = _operands(shape, gpu_device, seed=3)
        g = torch.Generator().manual_seed(1)
        noise = torch.randn(shape[3], shape[4], generator=g).to(gpu_device); bias = torch.randn(shape[2], generator=g).to(gpu_device)
        ys = {k: _mc()(x, w, s, d, noise, 0.7, bias, 3, 0.2, math.sqrt(2), 1.5, arith=c) for k, c in ARITH.items()}
        assert float(ys['fp32'].abs().max()) <= 1.5 and float((ys['fp32'].abs() == 1.5).float().mean()) > 0.01, 'the clamp must be active'
        torch.testing.assert_close(ys['bf16x6'], ys['fp32'], rtol=0, atol=2e-5)
        torch.testing.assert_close(ys['bf16x3'], ys['fp32'], rtol=0, atol=2e-4)


def test_default_arithmetic_switch(gpu_device):
    from torch_utils import hip_plugin
    before = hip_plugin.conv_arithmetic()
    try:
        x, w, s, d = _operands((1, 64, 64, 20, 20, 0), gpu_device)      # cin > 32: narrower layers always take the fp32 loop
        outs = {}
        for name, code in ARITH.items():
            assert hip_plugin.conv_arithmetic(name) == name
            outs[name] = _mc()(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0)
            assert torch.equal(outs[name], _mc()(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, arith=code)), 'process default != explicit arithmetic'
        assert not torch.equal(outs['fp32'], outs['bf16x3'])
        with pytest.raises(RuntimeError):
            hip_plugin.conv_arithmetic('tf32')
    finally:
        hip_plugin.conv_arithmetic('default')
    assert hip_plugin.conv_arithmetic() == before


@pytest.mark.parametrize('name', ['bf16x6', 'bf16x3'])
def test_reproducible_on_streams_and_in_graphs(gpu_device, name):
    code = ARITH[name]
    for shape in ((4, 128, 128, 128, 128, 0), (4, 128, 64, 128, 128, 2), (4, 512, 512, 16, 16, 0)):
        x, w, s, d = _operands(shape, gpu_device)
        call = lambda: _mc()(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=shape[5], arith=code)
        ref = call()
        assert all(torch.equal(ref, call()) for _ in range(3))
        st = torch.cuda.Stream(gpu_device)
        st.wait_stream(torch.cuda.current_stream(gpu_device))
        with torch.cuda.stream(st):
            call()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=st):
                yg = call()
        for _ in range(3):
            graph.replay()
            torch.cuda.synchronize(gpu_device)
            assert torch.equal(yg, ref)


def test_library_kernels_are_stable_beside_bf16_matrix_convolutions(gpu_device):
    """Victims: the style GEMVs, the head folding, both tri-plane gathers and bias_act (all VALU-heavy fp32 kernels) on stream B
    while stream A loops a bf16x6 convolution whose workgroups leave room on every CU.  With packed fp32 instructions in the
    victims this fails within a few launches (8-65 % of the launches were wrong)."""
    from torch_utils import hip_plugin
    from torch_utils.ops import bias_act
    dev = gpu_device
    g = torch.Generator().manual_seed(5)
    rn = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    x = rn(4, 64, 256, 256); wt = rn(64, 64, 3, 3); s = rn(4, 64) + 1; d = torch.rand(4, 64, generator=g).to(dev)
    conv = lambda: _mc()(x, wt, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, arith=6)
    w = rn(4, 512); A = rn(256, 512); bb = rn(256); wsq = torch.rand(256, 256, generator=g).to(dev)
    a1 = rn(128, 512); b1 = rn(128); w0 = rn(96, 128, 1, 1); w1 = rn(96, 128, 1, 1)
    planes = rn(4, 96, 256, 256).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)        # channels-last, like the backbone writes them
    coords = (torch.rand(4, 64 * 64 * 24, 3, generator=g) * 2 - 1).to(dev)
    xb = rn(4, 64, 128, 128); bias = rn(64)
    victims = {
        'style_demod': lambda: hip_plugin.StylePlugin.style_demod(w, A, bb, 1 / math.sqrt(512), 1.0, wsq),
        'fold_heads': lambda: hip_plugin.StylePlugin.fold_heads(w, 1 / math.sqrt(512), a1, b1, w0, 0.1, a1, b1, w1, 0.1),
        'triplane_sample': lambda: hip_plugin.TriplanePlugin.sample(planes, coords),
        'triplane_sample_rays': lambda: hip_plugin.TriplanePlugin.sample(planes, coords, ray_grid=(64, 64, 24)),
        'bias_act': lambda: bias_act.bias_act(xb, bias, act='lrelu'),
    }
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    with torch.cuda.stream(sa):
        conv()
    torch.cuda.synchronize(dev)
    flat = lambda o: [o] if torch.is_tensor(o) else [t for t in o if torch.is_tensor(t)]
    for name, fn in victims.items():
        with torch.cuda.stream(sb):
            first = fn()
        if first is None:
            continue
        torch.cuda.synchronize(dev)
        ref = [t.clone() for t in flat(first)]
        with torch.cuda.stream(sa):
            for _ in range(30):
                conv()
        with torch.cuda.stream(sb):
            outs = [flat(fn()) for _ in range(40 if name.startswith('triplane') else 200)]
        torch.cuda.synchronize(dev)
        bad = sum(1 for o in outs if not all(torch.equal(u, v) for u, v in zip(o, ref)))
        assert bad == 0, f'{name}: {bad} of {len(outs)} launches changed their result beside a bf16 matrix-core convolution'


def test_graphed_renderer_is_deterministic_with_split_arithmetic(gpu_device):
    """The failure that exposed the interaction: mapping + synthesis replayed from one hipGraph, style kernels overlapping the
    convolutions on a side branch: every replay must reproduce the eager pass bit for bit, also after a replay with other inputs."""
    from torch_utils import hip_plugin
    from training import triplane
    hip_plugin.conv_arithmetic('bf16x6')
    try:
        torch.manual_seed(0)
        G = triplane.TriPlaneGenerator().eval().to(gpu_device)          # full size: the split-bf16 loops need >= 64 output channels
        B = 2
        z = torch.from_numpy(np.stack([np.random.RandomState(s).randn(G.z_dim) for s in range(B)])).to(gpu_device)
        cams = torch.cat([triplane.camera_label(y) for y in (0.0, 0.4)]).to(gpu_device)
        cond = triplane.conditioning_label().repeat(B, 1).to(gpu_device)
        with torch.no_grad():
            ws = G.mapping(z.float(), cond)
            img_e, seg_e = G.synthesis(ws, c=cams, noise_mode='const', ray_jitter=False, return_seg=True)
        run = triplane.GraphedRenderer(G, B, gpu_device, ray_jitter=False)
        run(torch.randn(B, G.z_dim, device=gpu_device), cond, cams.flip(0))
        for _ in range(6):
            img, seg = run(z, cond, cams)
            assert torch.equal(img, img_e) and torch.equal(seg, seg_e)
    finally:
        hip_plugin.conv_arithmetic('default')


@pytest.mark.parametrize('shape', [(2, 64, 22, 200, 180), (4, 128, 192, 128, 130), (1, 100, 19, 255, 257), (2, 256, 192, 32, 32)], ids=lambda s: 'x'.join(map(str, s)))
def test_per_image_heads_vs_float64(gpu_device, shape):
    """The dual toRGB + toSeg heads (per-image folded 1x1 weights, bias, clamp) on the split-bf16 head kernel (>= 512 workgroups of 128 pixels, <= 32 or
    161..192 outputs; the last shape stays on the fp32 loop) against a float64 einsum; ragged pixel counts and a channel count that is no multiple of 16."""
    n, cin, cout, h, w = shape
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, cin, h, w, generator=g).to(gpu_device)
    wt = (torch.randn(n, cout, cin, 1, 1, generator=g) / math.sqrt(cin)).to(gpu_device)
    bias = torch.randn(cout, generator=g).to(gpu_device)
    ref = (torch.einsum('noc,nchw->nohw', wt[:, :, :, 0, 0].double(), x.double()) + bias.double()[None, :, None, None]).clamp(-2.0, 2.0)
    assert float((ref.abs() == 2.0).float().mean()) > 1e-4, 'the clamp must be active'
    for name, code in ARITH.items():
        y = _mc()(x, wt, None, None, None, 0.0, bias, 1, 0.0, 1.0, 2.0, arith=code).double()
        err = float((y - ref).abs().max() / ref.abs().max())
        assert err < TOL[name], f'{name}: {err:.3e}'
