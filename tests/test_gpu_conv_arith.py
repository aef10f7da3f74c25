"""The four arithmetics of the shared-weight 3x3 layers (fp32 MFMA / bf16x6 / f16x3 / bf16x3, include/ide3d_hip.h) and their
coexistence with the rest of the library on one GPU.

 * accuracy: each arithmetic against a float64 convolution (ATen on the GPU) on ragged and full-size shapes.  Stated tolerances
   (relative to max |ref|): fp32, bf16x6 and f16x3 4e-6 (fp32-grade: exact / <= 2^-23 / <= ~2^-21 products, fp32 accumulation),
   bf16x3 3e-5 (products to ~2^-17);
 * bf16x6 and f16x3 must be no worse than 3x the fp32 MFMA's own rounding error, rms;
 * adversarial operands (round 3): channel scales spread over 2^+-30, clamp-sized activations x 2^-20 weights, fp32 denormals,
   values beyond the largest bf16, inf / NaN — what every arithmetic guarantees there, and the documented differences;
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

ARITH = {'fp32': 1, 'bf16x6': 6, 'f16x3': 16, 'bf16x3': 3}
TOL = {'fp32': 4e-6, 'bf16x6': 4e-6, 'f16x3': 4e-6, 'bf16x3': 3e-5}
SHAPES = [  # n, cin, cout, h, w, mode
    (2, 40, 72, 37, 45, 0), (3, 64, 200, 33, 20, 0), (2, 512, 64, 16, 16, 0), (4, 128, 128, 256, 256, 0), (1, 17, 130, 12, 300, 0),
    (2, 40, 72, 37, 45, 2), (2, 96, 130, 20, 33, 2), (4, 512, 512, 16, 16, 2), (4, 128, 64, 256, 256, 2), (1, 33, 64, 13, 12, 2),
    (4, 48, 64, 250, 300, 0),      # 64 rows on 32 x 16-pixel tiles (round 4: >= 512 such tiles), ragged on both edges, 3 K chunks
    (3, 32, 128, 40, 70, 2),       # 2 K chunks on the split loop (transposed layers from 32 input channels on, round 4)
    (4, 512, 512, 4, 4, 0), (3, 256, 130, 7, 9, 0),      # small maps of wide layers: 8 x 16-pixel split tile, split-K 16 / 8 (round 4)
]


def _mc():
    """modconv2d with the bound on |x| that the f16x3 arithmetic needs (on the render path the producing kernel records it)."""
    from torch_utils import hip_plugin

    def call(x, *args, **kw):
        if kw.get('arith', 0) == 16 and 'x_amax' not in kw:
            kw['x_amax'] = _finite_amax(x)
        return hip_plugin.ModconvPlugin.modconv2d(x, *args, **kw)
    return call


def _finite_amax(x, slots=True):
    """max |finite x| per image; slots=True: in the [n, AMAX_FLOATS] layout of the C ABI (row maximum = the bound)."""
    a = x.abs()
    m = torch.where(torch.isfinite(a), a, torch.zeros_like(a)).amax(dim=(1, 2, 3))
    if not slots:
        return m
    out = torch.zeros(x.shape[0], 32 * 64, device=x.device)
    out[:, 5 * 64] = m
    return out


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
    assert rms['f16x3'] < 3 * rms['fp32'], f'f16x3 is not fp32-grade here: rms {rms}'


def test_epilogue_and_split_k_agree_between_arithmetics(gpu_device):
    """noise + bias + lrelu + gain + clamp on the split-bf16 loop (and its split-K reduction at 16x16) against the fp32 loop."""
    for shape in ((4, 512, 512, 16, 16, 0), (2, 64, 64, 40, 52, 0)):
        x, w, s, d = _operands(shape, gpu_device, seed=3)
        g = torch.Generator().manual_seed(1)
        noise = torch.randn(shape[3], shape[4], generator=g).to(gpu_device); bias = torch.randn(shape[2], generator=g).to(gpu_device)
        ys = {k: _mc()(x, w, s, d, noise, 0.7, bias, 3, 0.2, math.sqrt(2), 1.5, arith=c) for k, c in ARITH.items()}
        assert float(ys['fp32'].abs().max()) <= 1.5 and float((ys['fp32'].abs() == 1.5).float().mean()) > 0.01, 'the clamp must be active'
        torch.testing.assert_close(ys['bf16x6'], ys['fp32'], rtol=0, atol=2e-5)
        torch.testing.assert_close(ys['f16x3'], ys['fp32'], rtol=0, atol=2e-5)
        torch.testing.assert_close(ys['bf16x3'], ys['fp32'], rtol=0, atol=2e-4)


def _env_default_only():
    import os
    if os.environ.get('IDE3D_CONV_ARITH'):
        pytest.skip('asserts the library default (bf16x6); IDE3D_CONV_ARITH overrides it in this process')


def test_default_arithmetic_switch(gpu_device):
    _env_default_only()
    from torch_utils import hip_plugin
    before = hip_plugin.conv_arithmetic()
    try:
        x, w, s, d = _operands((1, 64, 64, 20, 20, 0), gpu_device)      # cin > 32: narrower layers always take the fp32 loop
        outs = {}
        assert before == 'bf16x6', 'the process default is the fp32-grade bf16x6 arithmetic (exclusive residency makes it safe beside foreign kernels)'
        for name, code in ARITH.items():
            assert hip_plugin.conv_arithmetic(name) == name
            amax = {'x_amax': _finite_amax(x)} if name == 'f16x3' else {}
            outs[name] = _mc()(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, **amax)
            assert torch.equal(outs[name], _mc()(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, arith=code)), 'process default != explicit arithmetic'
        assert not torch.equal(outs['fp32'], outs['bf16x3']) and not torch.equal(outs['f16x3'], outs['bf16x6'])
        # f16x3 without a bound on |x| runs in bf16x6 (bit for bit)
        assert torch.equal(hip_plugin.ModconvPlugin.modconv2d(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, arith=16), outs['bf16x6'])
        with pytest.raises(RuntimeError):
            hip_plugin.conv_arithmetic('tf32')
    finally:
        hip_plugin.conv_arithmetic('default')
    assert hip_plugin.conv_arithmetic() == before


@pytest.mark.parametrize('name', ['bf16x6', 'bf16x3', 'f16x3'])
def test_reproducible_on_streams_and_in_graphs(gpu_device, name):
    code = ARITH[name]
    for shape in ((4, 128, 128, 128, 128, 0), (4, 128, 64, 128, 128, 2), (4, 512, 512, 16, 16, 0)):
        x, w, s, d = _operands(shape, gpu_device)
        amax = _finite_amax(x)
        call = lambda: _mc()(x, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=shape[5], arith=code, x_amax=amax)
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
            for _ in range(400 if name.startswith('triplane') else 30):
                conv()
        with torch.cuda.stream(sb):
            outs = [flat(fn()) for _ in range(1000 if name.startswith('triplane') else 200)]      # the gathers failed most often in round 2
        torch.cuda.synchronize(dev)
        bad = sum(1 for o in outs if not all(torch.equal(u, v) for u, v in zip(o, ref)))
        assert bad == 0, f'{name}: {bad} of {len(outs)} launches changed their result beside a bf16 matrix-core convolution'


@pytest.mark.parametrize('arith', ['bf16x6', 'f16x3'])
def test_graphed_renderer_is_deterministic_with_split_arithmetic(gpu_device, arith):
    """The failure that exposed the interaction: mapping + synthesis replayed from one hipGraph, style kernels overlapping the
    convolutions on a side branch: every replay must reproduce the eager pass bit for bit, also after a replay with other inputs.
    (f16x3: the per-image amax buffers are zeroed and re-accumulated with atomics inside the graph — max is order-independent.)"""
    from torch_utils import hip_plugin
    from training import triplane
    hip_plugin.conv_arithmetic(arith)
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


@pytest.mark.parametrize('shape', [(2, 64, 22, 200, 180), (4, 128, 192, 128, 130), (1, 100, 19, 255, 257), (2, 256, 192, 32, 32), (2, 512, 192, 64, 64),
                                   (4, 512, 192, 8, 8), (3, 101, 19, 5, 7), (4, 512, 192, 16, 16), (2, 7, 5, 20, 20), (2, 512, 190, 30, 33)], ids=lambda s: 'x'.join(map(str, s)))
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


@pytest.mark.parametrize('shape', [(4, 128, 192, 256, 256), (4, 64, 22, 256, 256), (4, 128, 22, 256, 256), (3, 128, 161, 256, 192), (1, 64, 3, 512, 512), (5, 128, 19, 256, 256)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_resident_weight_heads_vs_float64(gpu_device, shape):
    """Round 4: heads whose packed weights fit LDS whole (K = 64 / 128) and whose images give every wave >= 2 tiles of 32 pixels run on
    head_resident_kernel (barrier-free grid-stride waves, direct accumulator stores): against a float64 einsum, with the clamp active, an
    odd number of images (85 / 51 workgroups per image), one image, unused rows (cout 161 / 19 / 3) and the recorded amax."""
    n, cin, cout, h, w = shape
    g = torch.Generator().manual_seed(19)
    x = torch.randn(n, cin, h, w, generator=g).to(gpu_device)
    wt = (torch.randn(n, cout, cin, 1, 1, generator=g) / math.sqrt(cin)).to(gpu_device)
    bias = torch.randn(cout, generator=g).to(gpu_device)
    ref = ((torch.einsum('noc,nchw->nohw', wt[:, :, :, 0, 0].double(), x.double()) + bias.double()[None, :, None, None]) * 0.75).clamp(-2.0, 2.0)
    assert float((ref.abs() == 2.0).float().mean()) > 1e-5, 'the clamp must be active'
    for name in ('bf16x6', 'bf16x3', 'f16x3'):
        amax = torch.zeros(n, 32 * 64, device=gpu_device)
        y = _mc()(x, wt, None, None, None, 0.0, bias, 1, 0.0, 0.75, 2.0, arith=ARITH[name], y_amax=amax)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        assert err < TOL[name if name != 'f16x3' else 'bf16x6'], f'{name}: {err:.3e}'          # f16x3: the heads stay on bf16x6
        assert torch.equal(amax.amax(dim=1), y.abs().amax(dim=(1, 2, 3)))
        y2 = _mc()(x, wt, None, None, None, 0.0, bias, 1, 0.0, 0.75, 2.0, arith=ARITH[name])
        assert torch.equal(y, y2)


# ---- round 3: the producers' amax, the f16x3 scales, adversarial operands ------------------------------------------------------------

@pytest.mark.parametrize('shape', [(2, 40, 72, 37, 45, 0), (4, 512, 512, 16, 16, 0), (3, 64, 64, 40, 52, 2), (2, 16, 24, 9, 7, 0), (5, 8, 40, 4, 4, 0)],
                         ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('name', ['fp32', 'bf16x6', 'f16x3'])
def test_y_amax_is_the_maximum_of_the_finished_output(gpu_device, shape, name):
    """`y_amax` (what the consuming f16x3 convolution scales its patch with) == max |y| per image, exactly, on every epilogue: vector /
    scalar stores, split-K reduction kernel, several images per tile; non-finite outputs are skipped."""
    x, w, s, d = _operands(shape, gpu_device, seed=11)
    g = torch.Generator().manual_seed(2)
    bias = torch.randn(shape[2], generator=g).to(gpu_device)
    amax = torch.zeros(shape[0], 32 * 64, device=gpu_device)
    y = _mc()(x, w, s, d, None, 0.0, bias, 3, 0.2, math.sqrt(2), -1.0, mode=shape[5], arith=ARITH[name], y_amax=amax)
    assert torch.equal(amax.amax(dim=1), y.abs().amax(dim=(1, 2, 3)))
    # NaNs are ignored by the running maximum (outputs that are NaN: linear epilogue; the branch-free lrelu / clamp stage turns a NaN
    # into -inf, like fmaxf in the reference's bias_act.cu does when a clamp is set) ...
    x2 = x.clone(); x2[-1, 1, 2, 2] = float('nan')
    amax.zero_()
    y = _mc()(x2, w, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=shape[5], arith=ARITH[name], y_amax=amax)
    fin = torch.where(torch.isnan(y), torch.zeros_like(y), y.abs())
    assert (~torch.isfinite(y)).any() and torch.equal(amax.amax(dim=1), fin.amax(dim=(1, 2, 3)))
    if name == 'fp32':                                              # ... an inf is kept (the split arithmetics turn it into NaN: inf - inf)
        x2[0, 0, 1, 1] = float('inf')
        amax.zero_()
        y = _mc()(x2, w, s, d, None, 0.0, bias, 3, 0.2, math.sqrt(2), -1.0, mode=shape[5], arith=ARITH[name], y_amax=amax)
        assert torch.isinf(y[0]).any() and float(amax.amax(dim=1)[0]) == float('inf')


def test_fir_epilogue_records_amax(gpu_device):
    from torch_utils import hip_plugin
    from torch_utils.ops import upfirdn2d
    g = torch.Generator().manual_seed(4)
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).to(gpu_device)
    for n, c, h in ((3, 20, 37), (2, 64, 129)):
        x = torch.randn(n, c, h, h, generator=g).to(gpu_device)
        noise = torch.randn(h - 1, h - 1, generator=g).to(gpu_device); bias = torch.randn(c, generator=g).to(gpu_device)
        amax = torch.zeros(n, 32 * 64, device=gpu_device)
        y = hip_plugin.Upfirdn2dPlugin.upfirdn2d_ex(x, f, 1, 1, 1, 1, 1, 1, 1, 1, False, 4.0, noise=noise, noise_strength=0.5, bias=bias, act=3, alpha=0.2,
                                                  act_gain=math.sqrt(2), y_amax=amax)
        y0 = hip_plugin.Upfirdn2dPlugin.upfirdn2d_ex(x, f, 1, 1, 1, 1, 1, 1, 1, 1, False, 4.0, noise=noise, noise_strength=0.5, bias=bias, act=3, alpha=0.2,
                                                   act_gain=math.sqrt(2))
        assert torch.equal(y, y0) and torch.equal(amax.amax(dim=1), y.abs().amax(dim=(1, 2, 3)))
    # generic kernel (5 x 5 filter), channels-last
    x = torch.randn(2, 6, 19, 23, generator=g).to(gpu_device).contiguous(memory_format=torch.channels_last)
    f5 = torch.rand(5, 5, generator=g).to(gpu_device)
    amax = torch.zeros(2, 32 * 64, device=gpu_device)
    y = hip_plugin.Upfirdn2dPlugin.upfirdn2d_ex(x, f5, 1, 1, 1, 1, 2, 2, 2, 2, False, 1.0, y_amax=amax)
    assert torch.equal(amax.amax(dim=1), y.abs().amax(dim=(1, 2, 3)))


def _errs(y, ref):
    """(max error / max |ref|, worst per-output-channel max error / that channel's max |ref|)."""
    err = (y.double() - ref).abs()
    ch = err.amax(dim=(0, 2, 3)) / ref.abs().amax(dim=(0, 2, 3)).clamp_min(1e-300)
    return float(err.max() / ref.abs().max()), float(ch.max())


@pytest.mark.parametrize('mode', [0, 2])
def test_adversarial_channel_scales_over_sixty_octaves(gpu_device, mode):
    """Styles 2^k, k uniform in [-30, 30], per input channel; output channels 0..31 see only the 16 smallest-scale inputs (their other
    weights are zero), so they sit ~2^-40 below the image maximum.  fp32 and bf16x6 keep every product to <= 2^-23 relative: every output
    channel is accurate relative to ITSELF.  f16x3 places x * s inside the fp16 range with ONE power of two per image: accurate relative
    to the image maximum (same 4e-6 bound), while channels more than 2^17 below it lose relative precision — the documented difference."""
    n, cin, cout, h, w = 2, 128, 96, 40, 36
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, cin, h, w, generator=g).to(gpu_device)
    k = torch.randint(-30, 31, (cin,), generator=g)
    order = torch.argsort(k)
    k[order[:16]] = torch.arange(-30, -14)                                  # guarantee a spread: 16 channels at 2^-30 .. 2^-15
    s = (2.0 ** k.float()).to(gpu_device)[None].repeat(n, 1) * (1 + 0.25 * torch.rand(n, cin, generator=g).to(gpu_device))
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(gpu_device)
    small = torch.zeros(cin, dtype=torch.bool); small[order[:16]] = True
    wt[:32][:, ~small.to(gpu_device)] = 0
    d = torch.ones(n, cout, device=gpu_device)
    ref = _ref64(x, wt, s, d, mode)
    assert float(ref[:, :32].abs().max() / ref.abs().max()) < 2.0 ** -35
    for name in ('fp32', 'bf16x6', 'f16x3'):
        y = _mc()(x, wt, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=mode, arith=ARITH[name])
        e_glob, e_chan = _errs(y, ref)
        assert e_glob < 4e-6, f'{name}: {e_glob:.2e} of max |ref|'
        if name != 'f16x3':
            assert e_chan < 1e-5, f'{name}: worst channel {e_chan:.2e} of its own maximum'
        else:
            big = _errs(y[:, 32:], ref[:, 32:])[1]
            assert big < 1e-5, f'f16x3: channels near the image maximum must be fp32-grade ({big:.2e})'
            assert torch.isfinite(y).all()


def test_adversarial_clamp_sized_activations_and_tiny_weights(gpu_device):
    """fp16-block conditions (inversion/networks.py:1058-1060): activations pinned at the conv_clamp (+-256) with weights ~2^-20."""
    n, cin, cout, h, w = 2, 96, 128, 48, 40
    g = torch.Generator().manual_seed(22)
    x = (torch.randn(n, cin, h, w, generator=g) * 300).clamp(-256, 256).to(gpu_device)
    assert float((x.abs() == 256).float().mean()) > 0.3
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 2.0 ** -20).to(gpu_device)
    s = (torch.randn(n, cin, generator=g) + 1).to(gpu_device); d = torch.ones(n, cout, device=gpu_device)
    for mode in (0, 2):
        ref = _ref64(x, wt, s, d, mode)
        for name in ('fp32', 'bf16x6', 'f16x3'):
            e_glob, e_chan = _errs(_mc()(x, wt, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=mode, arith=ARITH[name]), ref)
            assert e_glob < 4e-6 and e_chan < 1e-5, f'{name} mode {mode}: {e_glob:.2e} / {e_chan:.2e}'


def test_adversarial_denormal_operands(gpu_device):
    """fp32 denormals among normal data, and a whole image of denormal-sized activations: results stay within the stated tolerance of a
    float64 convolution (denormal residual pieces may flush: they are >= 2^-8 / 2^-11 below their operand)."""
    n, cin, cout, h, w = 2, 64, 64, 32, 32
    g = torch.Generator().manual_seed(23)
    x = torch.randn(n, cin, h, w, generator=g)
    x[0, :, ::3, ::2] = 1e-40 * torch.randn(cin, 11, 16, generator=g)      # denormals sprinkled into image 0
    x[1] *= 1e-36                                                          # image 1: everything close to the denormal range
    x = x.to(gpu_device)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(gpu_device)
    s = torch.ones(n, cin, device=gpu_device); d = torch.ones(n, cout, device=gpu_device)
    ref = _ref64(x, wt, s, d, 0)
    for name in ('fp32', 'bf16x6', 'f16x3'):
        y = _mc()(x, wt, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, arith=ARITH[name]).double()
        for i in range(n):
            e = float((y[i] - ref[i]).abs().max() / ref[i].abs().max())
            assert e < (4e-6 if i == 0 else 2e-3), f'{name}, image {i}: {e:.2e}'


def test_adversarial_beyond_bf16_and_non_finite_operands(gpu_device):
    """(1) Finite fp32 values that round to inf in bf16 (> 3.396e38; the largest bf16 is 3.3895e38): the fp32 loop and f16x3 (whose scale comes from the finite maximum) stay
    finite and accurate; bf16x6 rounds the leading piece to inf — documented: operands >= 2^127.99 are outside its domain.
    (2) inf / NaN activations: every arithmetic marks exactly the outputs whose receptive field contains the element as non-finite
    (inf may become NaN in the split arithmetics: inf - inf in the residual) and leaves all others at their finite values."""
    n, cin, cout, h, w = 1, 64, 64, 24, 24
    g = torch.Generator().manual_seed(24)
    x = torch.randn(n, cin, h, w, generator=g).to(gpu_device)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(gpu_device)
    s = torch.ones(n, cin, device=gpu_device); d = torch.ones(n, cout, device=gpu_device)
    call = lambda xx, ww, name: _mc()(xx, ww, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, arith=ARITH[name])
    # (1) one huge activation, weights small enough for a finite result
    xh = x.clone(); xh[0, 3, 10, 10] = 3.4e38
    wsm = wt * 2.0 ** -30
    ref = _ref64(xh, wsm, s, d, 0)
    assert torch.isfinite(ref).all()
    for name in ('fp32', 'f16x3'):
        y = call(xh, wsm, name)
        assert torch.isfinite(y).all() and _errs(y, ref)[0] < 4e-6, name
    assert not torch.isfinite(call(xh, wsm, 'bf16x6')).all(), 'documented: the leading bf16 piece of 3.4e38 is inf'
    xok = x.clone(); xok[0, 3, 10, 10] = 3.38e38                            # the largest bf16 is 3.3895e38: below it bf16x6 is fine
    assert torch.isfinite(call(xok, wsm, 'bf16x6')).all() and _errs(call(xok, wsm, 'bf16x6'), _ref64(xok, wsm, s, d, 0))[0] < 4e-6
    # (2) non-finite activations
    xn = x.clone(); xn[0, 5, 4, 7] = float('inf'); xn[0, 9, 15, 3] = float('nan'); xn[0, 20, 20, 20] = float('-inf')
    touched = torch.zeros(h, w, dtype=torch.bool, device=gpu_device)
    for yy, xx in ((4, 7), (15, 3), (20, 20)):
        touched[max(yy - 1, 0):yy + 2, max(xx - 1, 0):xx + 2] = True
    ref = _ref64(torch.where(torch.isfinite(xn), xn, torch.zeros_like(xn)), wt, s, d, 0)
    for name in ('fp32', 'bf16x6', 'f16x3'):
        y = call(xn, wt, name)
        bad = ~torch.isfinite(y)
        assert torch.equal(bad.any(dim=1)[0], touched) and bool(bad[:, :, touched].all()), f'{name}: non-finite outputs != receptive fields of the non-finite inputs'
        e = float((y.double() - ref).abs()[:, :, ~touched].max() / ref.abs().max())
        assert e < 4e-6, f'{name}: finite outputs beside non-finite ones: {e:.2e}'


def test_foreign_aten_kernels_beside_the_convolutions(gpu_device):
    """kernels of OTHER libraries next to the convolutions.  Victims: ATen element-wise / reduction kernels on stream B while
    stream A loops a convolution, in every arithmetic.  Since round 4 the matrix loops of the split arithmetics keep foreign waves off
    their SIMDs (exclusive residency), so the victims must be bit-stable beside ALL of them; the counts are also
    written to gpurun_out/aten_victims.json.  (The susceptible foreign kernel is tests/native/pk_victim.hip: next test.)"""
    import json, os
    from torch_utils import hip_plugin
    dev = gpu_device
    g = torch.Generator().manual_seed(6)
    rn = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    x = rn(4, 64, 256, 256); wt = rn(64, 64, 3, 3); s = rn(4, 64) + 1; d = torch.rand(4, 64, generator=g).to(dev)
    xam = _finite_amax(x)
    a, b, c = rn(1 << 22), rn(1 << 22), rn(1 << 22)
    m1, m2 = rn(512, 512), rn(512, 512)
    victims = {
        'addcmul': lambda: torch.addcmul(a, b, c, value=0.5),
        'mul_add': lambda: a * b + c,
        'sum': lambda: (a * b).sum(),
        'softmax': lambda: torch.softmax(m1, dim=1),
        'matmul_fp32': lambda: m1 @ m2,
    }
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    record = {}
    for arith, code in (('fp32', 1), ('bf16x6', 6), ('f16x3', 16)):
        conv = lambda: _mc()(x, wt, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, arith=code, x_amax=xam)
        with torch.cuda.stream(sa):
            conv()
        torch.cuda.synchronize(dev)
        for name, fn in victims.items():
            with torch.cuda.stream(sb):
                ref = fn().clone()
            torch.cuda.synchronize(dev)
            with torch.cuda.stream(sa):
                for _ in range(60):
                    conv()
            with torch.cuda.stream(sb):
                outs = [fn() for _ in range(300)]
            torch.cuda.synchronize(dev)
            bad = sum(1 for o in outs if not torch.equal(o, ref))
            record[f'{arith}/{name}'] = {'launches': len(outs), 'changed': bad}
            assert bad == 0, f'{name}: {bad} of {len(outs)} ATen launches changed beside the {arith} convolution'
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, 'aten_victims.json'), 'w') as f:
        json.dump(record, f, indent=1)


def _pk_victim_lib():
    import ctypes, os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'native', '_bin', 'libpk_victim.so')
    if not os.path.isfile(path):
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import __graft_entry__
        path = __graft_entry__.build_test_natives()
    if not os.path.isfile(path):
        pytest.skip('tests/native/pk_victim.hip could not be built here (hipcc output above)')
    lib = ctypes.CDLL(path)
    lib.pk_victim_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.pk_aggressor_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


def test_foreign_packed_fp32_victim_beside_every_matrix_loop(gpu_device):
    """DESIGN.md section 4.2, round 4.  tests/native/pk_victim.hip is a FOREIGN kernel with the susceptible pattern (v_pk_fma_f32 on operands
    straight from global loads; built with packed fp32 ON, unlike the library).
      positive control   beside a stand-alone LDS-read + bf16 MFMA loop whose waves SHARE SIMDs with it, its results change in most launches
                         (if the control shows nothing on this box the test is skipped: nothing to protect against);
      the library        beside every kernel of this library that contains an LDS-fed bf16 / fp16 matrix loop - the 3x3 and transposed
                         3x3 convolutions in bf16x6 / f16x3 / bf16x3 in their 8-wave, whole-SIMD-claiming and low-resolution forms, the
                         per-image heads, the fused ray-marcher and sample_voxel with bf16x6 MLPs - every one of 1500 victim launches per
                         neighbour must equal the result computed alone, bit for bit: exclusive residency keeps foreign waves off the
                         SIMDs those loops run on."""
    from torch_utils import hip_plugin
    from training import triplane, volumetric_rendering as vr
    lib = _pk_victim_lib()
    dev = gpu_device
    g = torch.Generator().manual_seed(12)
    rn = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    K, BLOCKS, REPS = 512, 64, 1500
    A = rn(BLOCKS * 32, K); xv = rn(K)
    ref = torch.empty(BLOCKS * 32, device=dev)
    ys = torch.empty(REPS, BLOCKS * 32, device=dev)
    spin = torch.empty(1024 * 256, device=dev)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)

    def victim(out, stream):
        assert lib.pk_victim_launch(A.data_ptr(), xv.data_ptr(), out.data_ptr(), K, BLOCKS, stream.cuda_stream) == 0

    victim(ref, sb)
    torch.cuda.synchronize(dev)

    def changed_launches(neighbour, every=25):
        """REPS victim launches on stream B while `neighbour()` keeps stream A busy; number of launches whose result differs from `ref`."""
        ys.zero_()
        torch.cuda.synchronize(dev)
        for i in range(REPS):
            if i % every == 0:
                with torch.cuda.stream(sa):
                    neighbour()
            victim(ys[i], sb)
        torch.cuda.synchronize(dev)
        return int((ys != ref[None]).any(dim=1).sum())

    control = changed_launches(lambda: lib.pk_aggressor_launch(spin.data_ptr(), 1500, 1024, sa.cuda_stream), every=40)
    if control == 0:
        pytest.skip('the stand-alone aggressor does not disturb the packed-fp32 victim on this device: nothing to protect against')
    assert changed_launches(lambda: None) == 0, 'the victim must be stable on its own'

    record = {'positive_control_changed_launches': control, 'launches_per_neighbour': REPS}
    neighbours = {}
    # convolutions: (cin, cout, res, mode) covering the 8-wave 16 x 16 form, the 64-row 16 x 16 form, 8 x 16 with the whole SIMD claimed, the
    # transposed 8-wave forms, the 4 x 16 transposed form and a low-resolution split-K layer
    for cin, cout, res, mode in ((128, 128, 256, 0), (64, 64, 512, 0), (512, 512, 64, 0), (256, 128, 128, 2), (128, 64, 256, 2), (512, 256, 64, 2), (512, 512, 16, 2)):
        x = rn(4, cin, res, res); wt = rn(cout, cin, 3, 3); s = rn(4, cin) + 1; d = torch.rand(4, cout, generator=g).to(dev)
        xam = _finite_amax(x)
        for arith, code in (('bf16x6', 6), ('f16x3', 16), ('bf16x3', 3), ('fp32', 1)):      # fp32: the LDS-fed v_mfma_f32_32x32x2_f32 loop, which shares SIMDs
            neighbours[f'conv {cin}->{cout} @{res} mode {mode} [{arith}]'] = \
                (lambda x=x, wt=wt, s=s, d=d, xam=xam, mode=mode, code=code: _mc()(x, wt, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=mode, arith=code, x_amax=xam))
    for cin, cout, res in ((128, 192, 256), (64, 22, 512)):
        x = rn(4, cin, res, res); wt = rn(4, cout, cin, 1, 1); bz = rn(cout)
        neighbours[f'heads {cin}->{cout} @{res} [bf16x6]'] = (lambda x=x, wt=wt, bz=bz: _mc()(x, wt, None, None, None, 0.0, bz, 1, 0.0, 1.0, 256.0, arith=6))
    # fused ray-marcher / sample_voxel with bf16x6 MLPs (they follow the process arithmetic)
    torch.manual_seed(1)
    R = triplane.TriplaneRenderer(triplane.GeneratorSpec()).to(dev).eval()
    tex = rn(4, 96, 256, 256).contiguous(memory_format=torch.channels_last); geo = rn(4, 96, 256, 256).contiguous(memory_format=torch.channels_last)
    cam = torch.cat([triplane.camera_label(y, device=dev) for y in (-0.5, 0.0, 0.5, 0.25)])[:, :16].reshape(-1, 4, 4)
    try:
        hip_plugin.conv_arithmetic('bf16x6')
        with torch.no_grad():
            R(tex, geo, cam, jitter=False)                              # warm (workspaces, plugin init) outside the measured window
        neighbours['render_rays [bf16x6 MLPs]'] = lambda: R(tex, geo, cam, jitter=False)
        before = hip_plugin.CALLS.get('render_rays', 0)
        for name, fn in neighbours.items():
            with torch.no_grad():
                bad = changed_launches(fn)
            record[name] = bad
        assert hip_plugin.CALLS.get('render_rays', 0) > before
        # the fp32 forms of the ray-marcher (v_mfma_f32_16x16x4_f32 fed from LDS, two 4-wave workgroups per CU: SIMDs are shared)
        hip_plugin.conv_arithmetic('fp32')
        with torch.no_grad():
            R(tex, geo, cam, jitter=False)
            record['render_rays [fp32 MLPs]'] = changed_launches(lambda: R(tex, geo, cam, jitter=False))
    finally:
        hip_plugin.conv_arithmetic('default')
    import json, os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, 'pk_victim_beside_library.json'), 'w') as f:
        json.dump(record, f, indent=1)
    hit = {k: v for k, v in record.items() if k not in ('positive_control_changed_launches', 'launches_per_neighbour') and v}
    assert not hit, f'foreign packed-fp32 victim disturbed beside: {hit} (positive control: {control} of {REPS})'


def test_amax_is_dropped_when_the_activation_is_edited_in_place(gpu_device):
    """VERDICT r3 item 6a / ADVICE: the producer's max |y| rides on the tensor object; an in-place edit between producer and consumer
    (`x.mul_(4)`: a hook, a viewer) must make the consumer IGNORE it (stale under-bound -> fp16 overflow) and run bf16x6 instead.  The
    guard is (object, `_version`, data pointer).  Result of the consumer: bit-equal to the bf16x6 launch and within 4e-6 of float64."""
    from torch_utils import hip_plugin
    from training import networks
    torch.manual_seed(3)
    la = networks.SynthesisLayer(64, 64, w_dim=32, resolution=64).to(gpu_device).eval()
    lb = networks.SynthesisLayer(64, 64, w_dim=32, resolution=64).to(gpu_device).eval()
    x = torch.randn(2, 64, 64, 64, device=gpu_device)
    w = torch.randn(2, 32, device=gpu_device)
    try:
        hip_plugin.conv_arithmetic('f16x3')
        with torch.no_grad():
            y = la(x, w, noise_mode='const')
            assert networks._amax_of(y) is not None, 'the producer must hand its amax on in f16x3'
            assert torch.equal(networks._amax_of(y).amax(dim=1), y.abs().amax(dim=(1, 2, 3)))
            out_clean = lb(y, w, noise_mode='const')                       # consumer in f16x3
            y2 = y.clone(); y2._ide3d_amax = y._ide3d_amax                 # a copy is another object at another address: also not trusted
            assert networks._amax_of(y2) is None
            y.mul_(4)                                                      # amax now under-estimates |y| by 4x
            assert networks._amax_of(y) is None, 'a stale amax must not reach the consumer'
            out_edit = lb(y, w, noise_mode='const')
        hip_plugin.conv_arithmetic('bf16x6')
        with torch.no_grad():
            out_b6 = lb(y, w, noise_mode='const')
    finally:
        hip_plugin.conv_arithmetic('default')
    assert torch.equal(out_edit, out_b6), 'without a trusted amax the f16x3 request must run the bf16x6 loop'
    assert torch.isfinite(out_edit).all()
    # float64 reference of layer b on the edited input
    with torch.no_grad():
        styles = lb.affine(w).double()
        wt = lb.weight.double()
        ww = wt[None] * styles[:, None, :, None, None]
        d = (ww.square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt()
        conv = torch.nn.functional.conv2d((y.double() * styles[:, :, None, None]).cpu(), wt.cpu(), padding=1).to(gpu_device) * d[:, :, None, None]
        ref = torch.nn.functional.leaky_relu(conv + (lb.noise_const * lb.noise_strength).double() + lb.bias.double()[None, :, None, None], 0.2) * math.sqrt(2)
    err = float((out_edit.double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 4e-6, err


@pytest.mark.parametrize('shape', [(4, 128, 64, 64, 64), (2, 64, 96, 32, 48), (1, 72, 64, 16, 128), (3, 40, 128, 33, 40)], ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('name', ['fp32', 'bf16x6', 'f16x3'])
def test_transposed_strip_plan_equals_full_grid(gpu_device, shape, name):
    """Round 4: the all-class transposed convolution on the h x w position grid + `tconv_strip_kernel` for output row 2h / column 2w
    (plain fp32 FMAs) against the same launch on the (h + 1) x (w + 1) grid (IDE3D_MODCONV_NO_STRIP): every output outside the strip is
    bit-equal, the strip agrees to fp32 rounding and with the float64 reference; padded rows (`pad_rows`) and `y_amax` included."""
    import os
    from torch_utils import hip_plugin
    n, cin, cout, h, w_ = shape
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, cin, h, w_, generator=g).to(gpu_device); wt = (torch.randn(cout, cin, 3, 3, generator=g) / 20).to(gpu_device)
    s = (torch.randn(n, cin, generator=g) + 1).to(gpu_device); d = (torch.rand(n, cout, generator=g) + 0.5).to(gpu_device)
    xam = _finite_amax(x)

    def run():
        am = torch.zeros(n, 32 * 64, device=gpu_device)
        y = _mc()(x, wt, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=2, arith=ARITH[name], x_amax=xam, y_amax=am, pad_rows=True)
        return y, am

    try:
        os.environ['IDE3D_MODCONV_NO_STRIP'] = '1'
        y_full, am_full = run()
    finally:
        os.environ.pop('IDE3D_MODCONV_NO_STRIP', None)
    y_strip, am_strip = run()
    assert y_strip.shape == (n, cout, 2 * h + 1, 2 * w_ + 1)
    assert torch.equal(y_strip[:, :, :-1, :-1], y_full[:, :, :-1, :-1]), 'outputs outside the strip must not depend on the plan'
    ref = torch.nn.functional.conv_transpose2d((x.double() * s.double()[:, :, None, None]).cpu(), wt.double().cpu().transpose(0, 1), stride=2) * d.double().cpu()[:, :, None, None]
    scale = float(ref.abs().max())
    for part, sl in (('row 2h', (slice(None), slice(None), -1, slice(None))), ('column 2w', (slice(None), slice(None), slice(None), -1))):
        e_ref = float((y_strip[sl].double().cpu() - ref[sl]).abs().max()) / scale
        e_full = float((y_strip[sl] - y_full[sl]).abs().max()) / scale
        assert e_ref < 4e-6 and e_full < (2e-5 if name == 'bf16x3' else 8e-6), f'{part}: vs float64 {e_ref:.2e}, vs the full-grid plan {e_full:.2e}'
    assert torch.equal(am_strip.amax(dim=1), y_strip.abs().amax(dim=(1, 2, 3))), 'y_amax must cover the strip'


@pytest.mark.parametrize('shape', [(2, 64, 128, 20, 24), (3, 80, 130, 16, 16), (3, 64, 384, 61, 70), (1, 48, 64, 33, 17), (4, 256, 128, 128, 128)],
                         ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('name', ['bf16x6', 'bf16x3', 'f16x3'])
def test_row_parity_pairs_equal_the_all_class_form(gpu_device, shape, name):
    """Round 5: `modconv_split_pair_kernel` (one output-row parity per workgroup: kernel rows {0, 2} / {1}, two accumulator sets on twice the
    positions) against the all-class form of the same layer (IDE3D_MODCONV_PAIR=0).  Every accumulator sees the same products in the same order,
    so wherever the two plans agree in blocking, split-K and strip the results (and `y_amax`) are BIT-EQUAL; otherwise both are within the
    arithmetic's tolerance of float64.  Ragged maps (tiles hanging over both edges), 64- and 128-row blocks, 2 .. 16 K chunks."""
    import os
    from torch_utils import hip_plugin
    n, cin, cout, h, w_ = shape
    g = torch.Generator().manual_seed(33)
    x = torch.randn(n, cin, h, w_, generator=g).to(gpu_device); wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).to(gpu_device)
    s = (torch.randn(n, cin, generator=g) + 1).to(gpu_device); d = (torch.rand(n, cout, generator=g) + 0.5).to(gpu_device)
    xam = _finite_amax(x)
    res = {}
    try:
        for knob in ('0', '2'):
            os.environ['IDE3D_MODCONV_PAIR'] = knob
            am = torch.zeros(n, 32 * 64, device=gpu_device)
            y = _mc()(x, wt, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=2, arith=ARITH[name], x_amax=xam, y_amax=am, pad_rows=True)
            pl = hip_plugin.modconv_plan(n, cin, cout, h, w_, mode=2, arith=ARITH[name], epilogue='plain', x_amax=True)
            res[knob] = (y, am.amax(dim=1), pl)
    finally:
        os.environ.pop('IDE3D_MODCONV_PAIR', None)
    (y0, a0, p0), (y2, a2, p2) = res['0'], res['2']
    assert p2['workgroups'] % 2 == 0 and (p2['tile_h'], p2['tile_w']) == ((16, 16) if p2['rows'] == 128 else (32, 16)) and p2['strip'] == 1 and p2['split_k'] == 1, p2
    assert (p0['tile_h'], p0['workgroups']) != (p2['tile_h'], p2['workgroups']), 'IDE3D_MODCONV_PAIR=0 must give the all-class form'
    ref = torch.nn.functional.conv_transpose2d(x.double() * s.double()[:, :, None, None], wt.double().transpose(0, 1), stride=2) * d.double()[:, :, None, None]
    scale = float(ref.abs().max())
    for y in (y0, y2):
        assert float((y.double() - ref).abs().max()) / scale < TOL[name]
    assert torch.equal(a2, y2.abs().amax(dim=(1, 2, 3)))
    if (p0['rows'], p0['strip'], p0['split_k'], p0['parts'], p0['f16']) == (p2['rows'], p2['strip'], p2['split_k'], p2['parts'], p2['f16']):
        assert torch.equal(y0, y2) and torch.equal(a0, a2), 'same products in the same order: the pair form must reproduce the all-class form bit for bit'


def test_every_matrix_loop_is_alone_on_its_cu_on_this_device(gpu_device):
    """Exclusive residency is checked where the kernels RUN, not assumed from the source: every launch of a kernel with an
    LDS-fed bf16 / fp16 matrix loop first asks the runtime how many of its workgroups fit one CU of this device
    (hipOccupancyMaxActiveBlocksPerMultiprocessor, once per kernel and device) and is refused unless the answer is one.  A full-size batch-4
    pass in every split arithmetic reaches all of them (3x3, transposed, two-team, low-resolution split-K forms, both head kernels, the
    ray-marcher with bf16x6 MLPs): none may have been refused."""
    from torch_utils import hip_plugin
    from training import graph_cache, triplane
    torch.manual_seed(0)
    G = triplane.TriPlaneGenerator().eval().requires_grad_(False).to(gpu_device)
    ws = torch.randn(4, G.num_ws, G.w_dim, device=gpu_device)
    c = torch.cat([triplane.camera_label(y, device=gpu_device) for y in (-0.5, 0.0, 0.5, 0.25)])
    keep = hip_plugin.conv_arithmetic()
    try:
        for arith in ('bf16x6', 'f16x3', 'bf16x3'):
            hip_plugin.conv_arithmetic(arith)
            with torch.no_grad(), graph_cache.disabled():
                img, seg = G.synthesis(ws, c=c, noise_mode='const', return_seg=True)
                img1, _ = G.synthesis(ws[:1], c=c[:1], noise_mode='const', return_seg=True)          # batch 1 plans other forms (more split-K)
            assert torch.isfinite(img).all() and torch.isfinite(img1).all()
    finally:
        hip_plugin.conv_arithmetic(keep)
    torch.cuda.synchronize()
    assert hip_plugin.exclusive_violations() == (0, ''), hip_plugin.exclusive_violations()
