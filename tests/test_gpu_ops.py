"""Parity of the HIP kernels (through the C ABI, behind the reference call surfaces) against the reference-generated
golden vectors and the CPU oracle.  Needs a real MI355X: `pytest -m gpu`.

Tolerances (fp32 unless stated): element-wise ops 2e-6 abs/rel; FIR / interpolation sums 1e-5; compositing 2e-5;
fp16 / bf16 storage 2e-3 / 2e-2 relative; tri-plane tap indices and in-bounds masks bit-exact.
"""

import math
import os

import numpy as np
import pytest
import torch

from oracle import fast_ops
from oracle import ops as oracle_ops
from util import assert_close, filter_from, t

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _native_calls():
    from torch_utils import hip_plugin
    hip_plugin.CALLS.clear()
    yield hip_plugin.CALLS


def _calls(name):
    from torch_utils import hip_plugin
    return hip_plugin.CALLS.get(name, 0)


# ---- bias_act ------------------------------------------------------------------------------------------------

def test_bias_act_golden(golden, gpu_device):
    from torch_utils.ops import bias_act
    for cfg, a in golden('bias_act'):
        kw = {k: cfg[k] for k in ('alpha', 'gain', 'clamp') if k in cfg}
        b = None if cfg.get('nobias') else t(a['in_b'], gpu_device)
        y = bias_act.bias_act(t(a['in_x'], gpu_device), b, dim=cfg['dim'], act=cfg['act'], **kw)
        assert y.is_cuda
        assert_close(y, a['out_y'], rtol=2e-6, atol=2e-6, what=str(cfg))
    assert _calls('bias_act') >= 29


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-6), (torch.float16, 2e-3), (torch.bfloat16, 2e-2), (torch.float64, 2e-7)])   # alpha / gain / clamp are fp32 in the ABI (as in bias_act.cpp:32)
def test_bias_act_dtypes_layouts_and_tails(gpu_device, dtype, tol):
    from torch_utils.ops import bias_act
    g = torch.Generator().manual_seed(0)
    for shape, cl in (((3, 5, 17, 13), False), ((2, 8, 9, 7), True), ((1, 3, 1, 1), False), ((4, 64, 32, 32), False)):
        x = torch.randn(*shape, generator=g).to(dtype)
        b = torch.randn(shape[1], generator=g).to(dtype)
        xd = x.to(gpu_device)
        if cl:
            xd = xd.contiguous(memory_format=torch.channels_last)
        for act in ('lrelu', 'softplus', 'swish', 'linear'):
            y = bias_act.bias_act(xd, b.to(gpu_device), act=act, clamp=1.5)
            ref = oracle_ops.bias_act(x.double(), b.double(), act=act, clamp=1.5)
            assert y.stride() == xd.stride() and y.dtype == dtype
            assert_close(y.double(), ref, rtol=tol, atol=tol, what=f'{dtype} {shape} {act}')


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
def test_bias_act_plane_kernel_equals_the_indexed_kernel(gpu_device, dtype):
    """Round 5: the forward pass over whole bias planes (`bias_act_plane_kernel`: bias found once per workgroup, no 64-bit index division
    per vector) against the indexed kernel (IDE3D_BIAS_ACT_NO_PLANES=1, read per call): bit-equal for every activation, with / without
    bias and clamp, planes that are not a multiple of the workgroup's 1024 vectors, and the no-bias call (one plane = the tensor)."""
    import os
    from torch_utils.ops import bias_act
    g = torch.Generator().manual_seed(4)
    for shape in ((2, 6, 128, 128), (1, 3, 72, 100), (3, 4, 40, 40), (2, 2, 300, 52)):
        x = (torch.randn(*shape, generator=g) * 2).to(dtype).to(gpu_device)
        b = torch.randn(shape[1], generator=g).to(dtype).to(gpu_device)
        for act in ('lrelu', 'linear', 'relu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish'):
            for bias, clamp in ((b, None), (b, 0.7), (None, 0.7)):
                try:
                    os.environ['IDE3D_BIAS_ACT_NO_PLANES'] = '1'
                    ref = bias_act.bias_act(x, bias, act=act, clamp=clamp)
                finally:
                    os.environ.pop('IDE3D_BIAS_ACT_NO_PLANES', None)
                y = bias_act.bias_act(x, bias, act=act, clamp=clamp)
                assert torch.equal(y.view(torch.int16 if dtype != torch.float32 else torch.int32), ref.view(torch.int16 if dtype != torch.float32 else torch.int32)), (shape, act, clamp)
        y = bias_act.bias_act(x, b, act='lrelu')
        want = oracle_ops.bias_act(x.cpu().double(), b.cpu().double(), act='lrelu')
        assert_close(y.double(), want, rtol=2e-2 if dtype != torch.float32 else 2e-6, atol=2e-2 if dtype != torch.float32 else 2e-6, what=f'{dtype} {shape}')


def test_bias_act_gradients(gpu_device):
    """First and second order gradients of the HIP op == autograd through the PyTorch definition."""
    from torch_utils.ops import bias_act
    g = torch.Generator().manual_seed(1)
    for act in bias_act.activation_funcs:
        x = torch.randn(2, 4, 6, 5, generator=g, dtype=torch.float64)
        b = torch.randn(4, generator=g, dtype=torch.float64)
        outs = []
        for dev, impl in ((gpu_device, 'cuda'), ('cpu', 'ref')):
            xx = x.to(dev).requires_grad_(True); bb = b.to(dev).requires_grad_(True)
            # 'linear' keeps neither x nor y for backward (bias_act.py:22, ref=''), so the reference kernel cannot
            # zero the gradient of clamped elements; that quirk is preserved, hence no clamp in the linear case.
            y = bias_act.bias_act(xx, bb, act=act, gain=1.3, clamp=(None if act == 'linear' else 2.0), impl=impl)
            w = torch.cos(torch.arange(y.numel(), dtype=torch.float64, device=dev)).reshape(y.shape)
            gx, gb = torch.autograd.grad((y * w).sum(), [xx, bb], create_graph=True)
            spec = bias_act.activation_funcs[act]
            if spec.has_2nd_grad:
                ggx, = torch.autograd.grad((gx * w).sum(), [xx], allow_unused=True)
            else:
                ggx = None
            outs.append((y, gx, gb, ggx))
        for name, a_, b_ in zip(('y', 'dx', 'db', 'd2x'), outs[0], outs[1]):
            if a_ is None or b_ is None:
                continue
            assert_close(a_, b_, rtol=1e-6, atol=1e-6, what=f'{act} {name}')
    assert _calls('bias_act') > 20


# ---- upfirdn2d -------------------------------------------------------------------------------------------------

def test_upfirdn2d_golden(golden, gpu_device):
    from torch_utils.ops import upfirdn2d
    for cfg, a in golden('upfirdn2d'):
        if 'fspec' in cfg:
            kw = {k: v for k, v in cfg.items() if k != 'fspec'}
            f = filter_from(a['in_f'])
            y = upfirdn2d.upfirdn2d(t(a['in_x'], gpu_device), None if f is None else f.to(gpu_device), **kw)
            assert_close(y, a['out_y'], rtol=1e-5, atol=1e-5, what=str(cfg))
        elif 'helper' in cfg:
            y = getattr(upfirdn2d, cfg['helper'])(t(a['in_x'], gpu_device), t(a['in_f'], gpu_device))
            assert_close(y, a['out_y'], rtol=1e-5, atol=1e-6, what=cfg['helper'])
    assert _calls('upfirdn2d') >= 14


@pytest.mark.parametrize('shape,kw', [
    ((2, 3, 65, 65), dict(up=1, padding=[1, 1, 1, 1], gain=4)),            # after a transposed conv (odd width)
    ((2, 5, 32, 32), dict(up=2, padding=[2, 1, 2, 1], gain=4)),            # skip-image upsample
    ((1, 4, 70, 38), dict(up=2, padding=[3, 0, 1, 2], gain=1)),            # odd pads: shifted polyphase cells
    ((1, 4, 64, 48), dict(down=2, padding=[1, 1, 1, 1])),
    ((1, 2, 300, 200), dict(up=1, padding=[2, 1, 2, 1], flip_filter=True)),
    ((1, 2, 40, 300), dict(up=2, padding=[2, 1, 2, 1], flip_filter=True, gain=4)),
])
@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-5), (torch.float16, 4e-3)])
def test_upfirdn2d_tile_kernels(gpu_device, shape, kw, dtype, tol):
    """The LDS-tiled polyphase kernels (4x4 filter) against the oracle, including tile-edge and odd-size cases."""
    from torch_utils.ops import upfirdn2d
    g = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=g).to(dtype)
    f = torch.tensor([[1., 2., 3., 4.], [0.5, 3., 3., 1.], [2., 3., 5., 1.], [1., 3., 3., 7.]]) / 40     # asymmetric
    y = upfirdn2d.upfirdn2d(x.to(gpu_device), f.to(gpu_device), **kw)
    ref = fast_ops.upfirdn2d(x.float(), f, **kw)
    assert y.dtype == dtype
    assert_close(y.float(), ref, rtol=tol, atol=tol, what=f'{shape} {kw}')


def test_upfirdn2d_cell_kernel_equals_generic(gpu_device, monkeypatch):
    """Large 2-D filters with up = 2 x 2 / 4 x 4 (viz/renderer.py:360: 47 x 47 taps, up 4; the reference's `upfirdn2d_kernel_large`) run the
    cell kernel (csrc/upfirdn2d.hip: one thread per U x U outputs, filter in LDS): bit-equal to the generic kernel — same taps, same order —
    and equal to the oracle; odd filter sizes, shifted / cropping pads, flip, fp16, channels_last, the fused epilogue."""
    from torch_utils.ops import upfirdn2d
    g = torch.Generator().manual_seed(12)
    cases = [((1, 3, 128, 128), (47, 47), dict(up=4, padding=[25, 22, 25, 22], gain=16)),          # upsample2d(f47, up=4)
             ((2, 5, 37, 29), (12, 12), dict(up=2, padding=[6, 5, 6, 5], gain=4)),
             ((1, 4, 33, 50), (9, 7), dict(up=2, padding=[3, 4, 1, 7], flip_filter=True)),
             ((2, 2, 40, 24), (8, 16), dict(up=4, padding=[-3, 9, 11, -2], gain=2.5)),                # negative pads crop
             ((1, 3, 16, 16), (6, 6), dict(up=4, padding=[0, 0, 0, 0]))]
    for shape, (fh, fw), kw in cases:
        f = torch.randn(fh, fw, generator=g) / (fh * fw) ** 0.5
        for dtype, tol in ((torch.float32, 2e-5), (torch.float16, 4e-3)):
            for fmt in (torch.contiguous_format, torch.channels_last):
                x = torch.randn(*shape, generator=g).to(dtype).to(gpu_device).contiguous(memory_format=fmt)
                y = upfirdn2d.upfirdn2d(x, f.to(gpu_device), **kw)
                monkeypatch.setenv('IDE3D_FIR_NO_CELL', '1')
                y_gen = upfirdn2d.upfirdn2d(x, f.to(gpu_device), **kw)
                monkeypatch.delenv('IDE3D_FIR_NO_CELL')
                assert y.dtype == dtype and torch.equal(y, y_gen), f'{shape} {fh}x{fw} {kw} {dtype} {fmt}'
                assert_close(y.float(), fast_ops.upfirdn2d(x.float().cpu().contiguous(), f, **kw), rtol=tol, atol=tol, what=f'{shape} {fh}x{fw} {kw} {dtype}')
    # with the fused epilogue (skip add + noise + bias + lrelu + clamp)
    from torch_utils import hip_plugin
    x = torch.randn(2, 4, 20, 20, generator=g).to(gpu_device); f = (torch.randn(10, 10, generator=g) / 10).to(gpu_device)
    add = torch.randn(2, 4, 40, 40, generator=g).to(gpu_device); nz = torch.randn(40, 40, generator=g).to(gpu_device); b = torch.randn(4, generator=g).to(gpu_device)
    run = lambda: hip_plugin.Upfirdn2dPlugin.upfirdn2d_ex(x, f, 2, 2, 1, 1, 5, 4, 5, 4, False, 4.0, add=add, noise=nz, noise_strength=0.3, bias=b, act=3, alpha=0.2, act_gain=1.4, clamp=1.1)
    y = run()
    monkeypatch.setenv('IDE3D_FIR_NO_CELL', '1')
    assert torch.equal(y, run())


def test_upfirdn2d_views_never_read_past_their_storage(gpu_device):
    """the 16-byte staging path of the tile kernels reads whole aligned quads of a row, i.e. up to round_up(in_w, 4) - 1.  It
    may only run when that much of EVERY row is readable (`x_row_floats`, ABI 5): W-offset views whose last row ends the storage, H-strided
    views with an odd row count and widths that are not multiples of 4 take the scalar staging.  Results against the oracle in every case;
    the promise the binding makes is checked directly as well."""
    from torch_utils import hip_plugin
    from torch_utils.ops import upfirdn2d
    g = torch.Generator().manual_seed(8)
    f = torch.tensor([[1., 2., 3., 4.], [0.5, 3., 3., 1.], [2., 3., 5., 1.], [1., 3., 3., 7.]]) / 40
    base = torch.randn(2, 3, 37, 52, generator=g).to(gpu_device)          # the storage ends with the last row of the last plane
    cases = {
        'dense, in_w % 4 == 0': base,
        'W-offset view t[..., 4:]': base[..., 4:],
        'W-offset view t[..., 3:] (unaligned rows)': base[..., 3:],
        'narrow view t[..., :45] (in_w % 4 = 1, padding readable inside the row)': base[..., :45],
        'H-strided view, odd row count': base[:, :, ::2],
        'H-strided view of a 50-wide tensor (in_w % 4 = 2)': torch.randn(1, 2, 33, 50, generator=g).to(gpu_device)[:, :, ::2],
    }
    for name, x in cases.items():
        n_readable = hip_plugin._row_floats_readable(x)
        last_row = x.storage_offset() + sum((x.shape[i] - 1) * x.stride(i) for i in range(3))
        total = x.untyped_storage().nbytes() // 4
        assert n_readable == total - last_row and n_readable >= x.shape[3], name
        for kw in (dict(up=1, padding=[1, 1, 1, 1], gain=4), dict(up=1, padding=[2, 1, 2, 1], flip_filter=True)):
            y = upfirdn2d.upfirdn2d(x, f.to(gpu_device), **kw)
            ref = fast_ops.upfirdn2d(x.cpu().contiguous(), f, **kw)
            assert_close(y, ref, rtol=1e-5, atol=1e-5, what=f'{name} {kw}')
    # the padded-row output of the transposed convolution (the case the fast staging exists for): rows of 65 floats inside a pitch of 68
    pad = torch.randn(2, 4, 65, 68, generator=g).to(gpu_device)
    view = pad[..., :65]
    assert hip_plugin._row_floats_readable(view) >= 68
    y = upfirdn2d.upfirdn2d(view, f.to(gpu_device), up=1, padding=[1, 1, 1, 1], gain=4)
    assert_close(y, fast_ops.upfirdn2d(view.cpu().contiguous(), f, up=1, padding=[1, 1, 1, 1], gain=4), rtol=1e-5, atol=1e-5, what='padded rows')


def test_upfirdn2d_channels_last_and_f64(gpu_device):
    from torch_utils.ops import upfirdn2d
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 6, 20, 22, generator=g)
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    ref = fast_ops.upfirdn2d(x, f, up=2, padding=[2, 1, 2, 1], gain=4)
    y = upfirdn2d.upfirdn2d(x.to(gpu_device).contiguous(memory_format=torch.channels_last), f.to(gpu_device), up=2, padding=[2, 1, 2, 1], gain=4)
    assert y.is_contiguous(memory_format=torch.channels_last)
    assert_close(y, ref, rtol=1e-5, atol=1e-5, what='channels_last')
    y = upfirdn2d.upfirdn2d(x.double().to(gpu_device), f.to(gpu_device), up=2, padding=[2, 1, 2, 1], gain=4)
    assert_close(y, oracle_ops.upfirdn2d(x.double(), f, up=2, padding=[2, 1, 2, 1], gain=4), rtol=1e-7, atol=1e-7, what='f64')


def test_upfirdn2d_gradient(gpu_device):
    from torch_utils.ops import upfirdn2d
    g = torch.Generator().manual_seed(5)
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    for kw in (dict(up=2, padding=[2, 1, 2, 1], gain=4), dict(down=2, padding=[1, 1, 1, 1]), dict(up=1, padding=[1, 1, 1, 1], gain=4)):
        x = torch.randn(2, 3, 12, 10, generator=g)
        grads = []
        for dev in (gpu_device, 'cpu'):
            xx = x.to(dev).requires_grad_(True)
            y = upfirdn2d.upfirdn2d(xx, f.to(dev), **kw)
            w = torch.sin(torch.arange(y.numel(), dtype=torch.float32, device=dev)).reshape(y.shape)
            grads.append(torch.autograd.grad((y * w).sum(), xx)[0])
        assert_close(grads[0], grads[1], rtol=1e-5, atol=1e-5, what=f'dx {kw}')


# ---- filtered_lrelu ----------------------------------------------------------------------------------------------

def test_filtered_lrelu_golden(golden, gpu_device):
    from torch_utils.ops import filtered_lrelu
    for cfg, a in golden('filtered_lrelu'):
        fu, fd = filter_from(a['in_fu']), filter_from(a['in_fd'])
        y = filtered_lrelu.filtered_lrelu(t(a['in_x'], gpu_device), fu=None if fu is None else fu.to(gpu_device),
                                          fd=None if fd is None else fd.to(gpu_device), b=t(a['in_b'], gpu_device), **cfg)
        assert_close(y, a['out_y'], rtol=1e-5, atol=1e-5, what=str(cfg))
    assert _calls('filtered_lrelu') >= 5


def test_filtered_lrelu_signs_and_backward(gpu_device):
    """Sign tensor == oracle's 2-bit codes; backward through the sign tensor == autograd through the definition."""
    from torch_utils import hip_plugin
    from torch_utils.ops import filtered_lrelu, upfirdn2d
    g = torch.Generator().manual_seed(6)
    f = upfirdn2d.setup_filter([1, 2, 3, 4, 5, 6, 6, 5, 4, 3, 2, 1], separable=True)
    x = torch.randn(2, 3, 12, 11, generator=g)
    b = torch.randn(3, generator=g)
    kw = dict(up=2, down=2, padding=[9, 10, 9, 10], gain=1.3, slope=0.25, clamp=0.8)
    # raw plugin call to look at the sign tensor
    plug = hip_plugin.FilteredLReluPlugin
    y, so, rc = plug.filtered_lrelu(x.to(gpu_device), f.to(gpu_device), f.to(gpu_device), b.to(gpu_device), torch.empty([0]),
                                    2, 2, 9, 10, 9, 10, 0, 0, 1.3, 0.25, 0.8, False, True)
    assert rc == 0
    yo, codes = oracle_ops.filtered_lrelu(x, fu=f, fd=f, b=b, return_signs=True, **kw)
    assert_close(y, yo, rtol=1e-5, atol=1e-5, what='y with sign write')
    so = so.cpu().numpy()
    H, Wc = codes.shape[2], codes.shape[3]
    unpacked = np.stack([(so >> (2 * j)) & 3 for j in range(4)], axis=-1).reshape(so.shape[0], so.shape[1], so.shape[2], -1)
    mism = (unpacked[:, :, :H, :Wc] != codes.numpy())
    # codes can only differ where the intermediate is within rounding distance of 0 or of the clamp
    assert mism.mean() < 1e-3
    # gradients
    grads = []
    for dev, impl in ((gpu_device, 'cuda'), ('cpu', 'ref')):
        xx = x.to(dev).requires_grad_(True); bb = b.to(dev).requires_grad_(True)
        yy = filtered_lrelu.filtered_lrelu(xx, fu=f.to(dev), fd=f.to(dev), b=bb, impl=impl, **kw)
        w = torch.sin(torch.arange(yy.numel(), dtype=torch.float32, device=dev)).reshape(yy.shape)
        grads.append(torch.autograd.grad((yy * w).sum(), [xx, bb]))
    assert_close(grads[0][0], grads[1][0], rtol=1e-4, atol=1e-4, what='dx')
    assert_close(grads[0][1], grads[1][1], rtol=1e-4, atol=1e-3, what='db')


def test_filtered_lrelu_sign_bytes_exact(gpu_device):
    """The packed sign tensor is BYTE output: bit-exact, not "within tolerance".  Inputs are built so that every intermediate is
    exactly representable (dyadic data, bias, filter taps, gain, slope and clamp: every fp32 product and partial sum is exact in
    any order), hence the up-sampled value that is classified is bit-identical on the device and in the fp64 oracle, including
    exact zeros and values exactly at the clamp.  Format: 4 x 2-bit codes per byte, element 4k + j in bits 2j..2j+1, code bit 0 =
    negative, bit 1 = clamped (filtered_lrelu.cu:494-519); rows padded to a multiple of 16 elements (filtered_lrelu.cpp:89-93)."""
    from torch_utils import hip_plugin
    from torch_utils.ops import upfirdn2d
    plug = hip_plugin.FilteredLReluPlugin
    g = torch.Generator().manual_seed(16)
    cases = (
        # 2-D 4x4 filter [1,3,3,1] (x) [1,3,3,1] / 64, up 2 / down 2
        (upfirdn2d.setup_filter([1, 3, 3, 1]), (2, 3, 13, 10), dict(up=2, down=2, padding=[2, 1, 2, 1], gain=2.0, slope=0.25, clamp=0.75)),
        # separable 12-tap filter whose taps sum to 64, up 2 / down 2 (StyleGAN3 layer shape)
        (upfirdn2d.setup_filter([1, 1, 2, 4, 8, 16, 16, 8, 4, 2, 1, 1], separable=True), (2, 4, 17, 19),
         dict(up=2, down=2, padding=[10, 11, 10, 11], gain=2.0, slope=0.125, clamp=0.5)),
        # up 4 / down 2, no clamp
        (upfirdn2d.setup_filter([1, 1, 2, 4, 8, 16, 16, 8, 4, 2, 1, 1], separable=True), (1, 2, 9, 12),
         dict(up=4, down=2, padding=[10, 11, 10, 11], gain=4.0, slope=0.5, clamp=None)),
    )
    for f, shape, kw in cases:
        x = torch.randint(-6, 7, shape, generator=g).float() / 4
        b = torch.randint(-4, 5, (shape[1],), generator=g).float() / 8
        p = kw['padding']
        clamp = kw['clamp'] if kw['clamp'] is not None else float('inf')
        y, so, rc = plug.filtered_lrelu(x.to(gpu_device), f.to(gpu_device), f.to(gpu_device), b.to(gpu_device), torch.empty([0]),
                                        kw['up'], kw['down'], p[0], p[1], p[2], p[3], 0, 0, kw['gain'], kw['slope'], clamp, False, True)
        assert rc == 0
        yo, codes = oracle_ops.filtered_lrelu(x, fu=f, fd=f, b=b, return_signs=True, **kw)
        codes = codes.numpy()
        frac = [float((codes == k).mean()) for k in range(3)]
        assert frac[0] > 0.05 and frac[1] > 0.05 and (kw['clamp'] is None or frac[2] > 0.05), f'degenerate test data: code frequencies {frac}'
        # the classified (up-sampled) values need <= 18 mantissa bits in every case; the final output only for the 4x4 filter (the
        # two 1-D down-sampling passes of the separable filter add 12 more bits), so only there it must be bit-equal as well
        if f.ndim == 2:
            assert torch.equal(y.cpu(), yo.float()), 'exactly representable data: the output itself must be bit-equal too'
        else:
            assert_close(y, yo, rtol=1e-5, atol=1e-5, what='y')
        so = so.cpu().numpy()
        n, c, H, Wc = codes.shape
        assert so.dtype == np.uint8 and so.shape[:2] == (n, c) and so.shape[2] >= H and so.shape[3] * 4 >= Wc and (so.shape[3] * 4) % 16 == 0
        full = Wc // 4
        packed = np.zeros((n, c, H, (Wc + 3) // 4), dtype=np.uint8)
        for j in range(4):
            col = codes[:, :, :, j::4].astype(np.uint8) << (2 * j)
            packed[:, :, :, :col.shape[3]] |= col
        assert np.array_equal(so[:, :, :H, :full], packed[:, :, :, :full]), f'packed sign bytes differ: {kw}'
        if Wc % 4:                          # last, partially valid byte of a row: compare the valid 2-bit fields only
            mask = np.uint8((1 << (2 * (Wc % 4))) - 1)
            assert np.array_equal(so[:, :, :H, full] & mask, packed[:, :, :, full] & mask)
        # reading the signs back (backward-pass mode of the same kernel) reproduces gain * slope / 0 / gain per element
        dy = torch.randint(-4, 5, x.shape, generator=g).float() / 4
        dx, _so2, rc = plug.filtered_lrelu(dy.to(gpu_device), f.to(gpu_device), f.to(gpu_device), torch.zeros(shape[1], device=gpu_device),
                                           plug.filtered_lrelu(x.to(gpu_device), f.to(gpu_device), f.to(gpu_device), b.to(gpu_device), torch.empty([0]),
                                                               kw['up'], kw['down'], p[0], p[1], p[2], p[3], 0, 0, kw['gain'], kw['slope'], clamp, False, True)[1],
                                           kw['up'], kw['down'], p[0], p[1], p[2], p[3], 0, 0, kw['gain'], kw['slope'], clamp, False, False)
        assert rc == 0
        z = oracle_ops.upfirdn2d(dy.double(), f, up=kw['up'], padding=p, gain=kw['up'] ** 2).numpy()
        scale = np.where(codes == 2, 0.0, np.where(codes == 1, kw['gain'] * kw['slope'], kw['gain']))
        want = oracle_ops.upfirdn2d(torch.from_numpy(z * scale), f, down=kw['down']).float()
        if f.ndim == 2:
            assert torch.equal(dx.cpu(), want), 'sign-read mode'
        else:
            assert_close(dx, want, rtol=1e-5, atol=1e-5, what='sign-read mode')


def test_filtered_lrelu_generic_fallback_path(gpu_device):
    """`return_code = -1` route: upfirdn2d -> filtered_lrelu_act_ -> upfirdn2d (float64 has no fused kernel)."""
    from torch_utils.ops import filtered_lrelu, upfirdn2d
    g = torch.Generator().manual_seed(7)
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    x = torch.randn(1, 2, 9, 9, generator=g, dtype=torch.float64)
    b = torch.randn(2, generator=g, dtype=torch.float64)
    kw = dict(up=2, down=1, padding=[2, 1, 2, 1], clamp=0.7)
    with pytest.warns(RuntimeWarning):
        y = filtered_lrelu.filtered_lrelu(x.to(gpu_device), fu=f.to(gpu_device), fd=f.to(gpu_device), b=b.to(gpu_device), **kw)
    assert_close(y, oracle_ops.filtered_lrelu(x, fu=f, fd=f, b=b, **kw), rtol=1e-6, atol=1e-6, what='generic path')
    assert _calls('filtered_lrelu_act_') == 1


# ---- tri-plane gather ------------------------------------------------------------------------------------------------

def test_triplane_golden_and_layouts(golden, gpu_device):
    from dnnlib import util
    for cfg, a in golden('triplane'):
        grid, co = t(a['in_grid'], gpu_device), t(a['in_coords'], gpu_device)
        for g_ in (grid, grid.contiguous(memory_format=torch.channels_last)):
            out = util.sample_from_triplane(co, g_, ray_grid=False)
            assert_close(out, a['out_feat'], rtol=1e-5, atol=2e-6, what=str(cfg))
            if cfg.get('ray_grid'):          # round-6 cases: ray grids of the generator's channel count -> the LDS-staged kernel as well
                out = util.sample_from_triplane(co, g_, ray_grid=tuple(cfg['ray_grid']))
                assert_close(out, a['out_feat'], rtol=1e-5, atol=2e-6, what=str(cfg) + ' (tile kernel)')
    assert _calls('triplane_sample') == 10 and _calls('triplane_sample_rays') == 4


def test_triplane_tap_indices_bit_exact(gpu_device):
    """Integer tap origins and in-bounds masks equal the oracle's (== ATen's, tests/test_oracle_golden.py)."""
    from torch_utils import hip_plugin
    g = torch.Generator().manual_seed(8)
    co = (torch.rand(200000, 3, generator=g) * 2 - 1) * 1.05
    co[:8] = torch.tensor([[1, -1, 0], [-1, 1, 1], [0.999999, -0.999999, 0.5], [1.0000001, 0, 0], [-1.0000001, 0, 0],
                           [float('nan'), 0, 0], [float('inf'), 0, 0], [1e30, -1e30, 0]])
    for H in (256, 37):
        taps = hip_plugin.TriplanePlugin.taps(H, H, co.to(gpu_device)).cpu().numpy()
        ref = oracle_ops.triplane_taps(np.nan_to_num(co.numpy(), nan=1e9, posinf=1e9, neginf=-1e9), H, H)
        finite = np.isfinite(co.numpy()).all(axis=1)
        assert np.array_equal(taps[finite], ref[finite])
        assert (taps[~finite][:, 0, 2] == 0).all()          # non-finite coordinates never touch memory


def test_triplane_full_size_properties(gpu_device):
    """At the benchmark size (C=32, 256^2, M=393216): linearity in the planes and exactness on constant planes."""
    from dnnlib import util
    g = torch.Generator(device='cpu').manual_seed(9)
    n, C, H, M = 1, 32, 256, 64 * 64 * 96
    p1 = torch.randn(n, 3 * C, H, H, generator=g).to(gpu_device).contiguous(memory_format=torch.channels_last)
    p2 = torch.randn(n, 3 * C, H, H, generator=g).to(gpu_device).contiguous(memory_format=torch.channels_last)
    co = ((torch.rand(n, M, 3, generator=g) * 2 - 1) * 0.7).to(gpu_device)
    o1, o2, o12 = util.sample_from_triplane(co, p1), util.sample_from_triplane(co, p2), util.sample_from_triplane(co, p1 + 2 * p2)
    assert_close(o12, o1 + 2 * o2, rtol=1e-5, atol=2e-5, what='linearity')
    const = torch.ones_like(p1) * 0.25
    assert_close(util.sample_from_triplane(co, const), torch.full((M, C), 0.75), rtol=0, atol=1e-6, what='constant planes')
    # subset against the oracle
    idx = torch.arange(0, M, 997, device=gpu_device)
    ref = fast_ops.sample_from_triplane(co[:, idx].cpu(), p1.cpu().contiguous())
    assert_close(o1[idx], ref, rtol=1e-5, atol=2e-5, what='subset vs oracle')


def test_triplane_ray_grid_kernel_equals_flat_kernel(gpu_device):
    """`ide3d_triplane_sample_rays` (LDS-staged 8x8-ray tiles) == `ide3d_triplane_sample`, bit for bit: frustum coordinates
    (regions staged), uniformly random and wild coordinates (regions too large / out of the plane -> direct loads), ray
    grids the tile kernel does not cover (falls back to the flat kernel), and vs the oracle on a subset."""
    from dnnlib import util
    from training import triplane, volumetric_rendering as vr
    g = torch.Generator().manual_seed(21)
    n, C, H = 3, 32, 256
    planes = torch.randn(n, 3 * C, H, H, generator=g).to(gpu_device).contiguous(memory_format=torch.channels_last)

    def frustum(res, steps, yaws, pitch=math.pi / 2):
        pts, z, d = vr.get_initial_rays_trig(len(yaws), steps, gpu_device, 18.0, res, 2.25, 3.3)
        cam = torch.cat([triplane.camera_label(y, pitch=pitch, device=gpu_device) for y in yaws])[:, :16].reshape(-1, 4, 4)
        wp, *_ = vr.transform_sampled_points(pts, z, d, gpu_device, h_stddev=0, v_stddev=0, camera=cam, mode=None,
                                             jitter=torch.rand(z.shape, generator=g).to(gpu_device))
        return wp.reshape(len(yaws), -1, 3).contiguous()

    cases = []
    cases.append(('frustum 64x64x96', frustum((64, 64), 96, (-0.6, 0.0, 0.45)), (64, 64, 96)))
    cases.append(('frustum 16x24x12 pitched', frustum((16, 24), 12, (1.2, -1.5, 3.0), pitch=1.1), (16, 24, 12)))
    cases.append(('frustum x2.6 (leaves the planes)', frustum((32, 32), 48, (0.3, 0.9, -2.0)) * 2.6, (32, 32, 48)))
    rnd = (torch.rand(n, 16 * 16 * 8, 3, generator=g) * 2 - 1) * 1.1
    rnd[0, :6] = torch.tensor([[float('nan'), 0, 0], [float('inf'), 0.1, 0], [1e30, -1e30, 0], [1, -1, 1], [-1, 1, -1], [0.999999, 0, -0.999999]])
    cases.append(('random + wild', rnd.to(gpu_device), (16, 16, 8)))
    cases.append(('grid not covered by the tile kernel', frustum((12, 20), 10, (0.1, 0.2, 0.3)), (12, 20, 10)))
    for name, co, rg in cases:
        flat = util.sample_from_triplane(co, planes, ray_grid=False)
        tiled = util.sample_from_triplane(co, planes, ray_grid=rg)
        assert torch.equal(torch.nan_to_num(flat, nan=123.0), torch.nan_to_num(tiled, nan=123.0)), name
    assert _calls('triplane_sample_rays') == len(cases) and _calls('triplane_sample') == len(cases)
    # the call exactly as the reference spells it (dnnlib/util.py:580: no hint): the ray grid is recognised from the data on the first call with
    # this M (cached afterwards) and the LDS-staged kernel runs; coordinates that are no ray grid keep the flat kernel; results bit-equal
    util._ray_grid_cache.clear()
    for name, co, rg in cases[:3]:
        before = _calls('triplane_sample_rays')
        plain = util.sample_from_triplane(co, planes)
        assert _calls('triplane_sample_rays') == before + 1, name
        assert util._ray_grid_cache[(co.shape[1], gpu_device.index)] in (rg, (rg[1], rg[0], rg[2])), (name, util._ray_grid_cache)     # (rows, row length, steps)
        assert torch.equal(torch.nan_to_num(plain, nan=123.0), torch.nan_to_num(util.sample_from_triplane(co, planes, ray_grid=False), nan=123.0)), name
    util._ray_grid_cache.clear()
    before = _calls('triplane_sample')
    util.sample_from_triplane(cases[3][1], planes)
    assert _calls('triplane_sample') == before + 1 and util._ray_grid_cache[(cases[3][1].shape[1], gpu_device.index)] is None
    co = cases[0][1]
    idx = torch.arange(0, co.shape[1], 1009, device=gpu_device)
    ref = fast_ops.sample_from_triplane(co[:, idx].cpu(), planes.cpu().contiguous())
    got = util.sample_from_triplane(co, planes, ray_grid=(64, 64, 96)).reshape(n, -1, C)[:, idx].reshape(-1, C)
    assert_close(got, ref, rtol=1e-5, atol=2e-5, what='tiled vs oracle')


def test_triplane_backward(gpu_device):
    from dnnlib import util
    g = torch.Generator().manual_seed(10)
    grid = torch.randn(2, 3 * 8, 16, 16, generator=g)
    co = (torch.rand(2, 50, 3, generator=g) * 2 - 1) * 1.1
    res = []
    for dev in (gpu_device, 'cpu'):
        gg = grid.to(dev).requires_grad_(True); cc = co.to(dev).requires_grad_(True)
        out = util.sample_from_triplane(cc, gg)
        w = torch.sin(torch.arange(out.numel(), dtype=torch.float32, device=dev)).reshape(out.shape)
        res.append(torch.autograd.grad((out * w).sum(), [gg, cc]))
    assert_close(res[0][0], res[1][0], rtol=1e-5, atol=1e-5, what='d planes')
    assert_close(res[0][1], res[1][1], rtol=1e-4, atol=1e-4, what='d coords')
    assert _calls('triplane_sample_backward') == 1


# ---- compositing -----------------------------------------------------------------------------------------------------

def test_fancy_integration_golden(golden, gpu_device):
    from training import volumetric_rendering as vr
    cases = golden('volumetric').select(fn='fancy_integration')
    for cfg, a in cases:
        kw = {k: cfg[k] for k in ('clamp_mode', 'last_back', 'white_back', 'max_depth', 'fill_mode') if k in cfg}
        noise = t(a['in_noise'], gpu_device) if 'in_noise' in a else None
        with torch.no_grad():
            rgb, depth, w = vr.fancy_integration(t(a['in_rs'], gpu_device), t(a['in_d'], gpu_device), t(a['in_z'], gpu_device), gpu_device,
                                                 noise_std=cfg.get('noise_std', 0), noise=noise, **kw)
        assert_close(w, a['out_w'], rtol=2e-5, atol=1e-6, what=f'weights {cfg}')
        assert_close(rgb, a['out_rgb'], rtol=2e-5, atol=2e-5, what=f'rgb {cfg}')
        assert_close(depth, a['out_depth'], rtol=2e-5, atol=2e-5, what=f'depth {cfg}')
    assert _calls('composite') == len(cases)


def test_fancy_integration_96_steps_52_channels(gpu_device):
    from training import volumetric_rendering as vr
    g = torch.Generator().manual_seed(11)
    rs = torch.randn(2, 300, 96, 52, generator=g); rs[..., -1] *= 3
    z = torch.sort(torch.rand(2, 300, 96, 1, generator=g) * 1.05 + 2.25, dim=2)[0]
    d = torch.randn(2, 300, 3, generator=g)
    with torch.no_grad():
        rgb, depth, w = vr.fancy_integration(rs.to(gpu_device), d.to(gpu_device), z.to(gpu_device), gpu_device, noise_std=0, clamp_mode='softplus')
    r_rgb, r_depth, r_w = oracle_ops.composite(rs, d, z)
    assert_close(w, r_w, rtol=2e-5, atol=1e-6, what='weights')
    assert_close(rgb, r_rgb, rtol=2e-5, atol=2e-5, what='rgb')
    assert_close(depth, r_depth, rtol=2e-5, atol=2e-5, what='depth')
    # a fully opaque ray integrates to weight sum 1 (partition of unity)
    rs2 = rs.clone(); rs2[..., -1] = 50.0
    with torch.no_grad():
        _, _, w2 = vr.fancy_integration(rs2.to(gpu_device), d.to(gpu_device), z.to(gpu_device), gpu_device, noise_std=0, clamp_mode='relu')
    assert_close(w2.sum(2), torch.ones(2, 300, 1), rtol=0, atol=1e-5, what='weights sum to one')


# ---- modulated convolution (fp32 MFMA implicit GEMM) ---------------------------------------------------------------------

def test_sample_pdf_kernel(golden, gpu_device):
    """`ide3d_sample_pdf` against the reference-run vectors (det linspace; random draws with empty bins and an all-zero
    ray), against the oracle at the benchmark shape, and on degenerate shapes.  Tolerance 2e-6 relative to the depth
    range: the cdf is rounded from double prefix sums like ATen's CPU cumsum, the rest is the same float expression."""
    from training import volumetric_rendering as vr
    from torch_utils import hip_plugin
    hip_plugin.CALLS.clear()
    (cfg, a), = golden('volumetric').select(fn='sample_pdf')
    s = vr.sample_pdf(t(a['in_bins'], gpu_device), t(a['in_w'], gpu_device), cfg['N_importance'], det=True)
    assert_close(s, a['out_samples'], rtol=2e-6, atol=2e-6, what='sample_pdf det')
    (cfg, a), = golden('volumetric').select(fn='sample_pdf_rand')
    s = vr.sample_pdf(t(a['in_bins'], gpu_device), t(a['in_w'], gpu_device), cfg['N_importance'], u=t(a['in_u'], gpu_device))
    assert_close(s, a['out_samples'], rtol=2e-6, atol=2e-6, what='sample_pdf rand')
    assert _calls('sample_pdf') == 2
    # benchmark shape: 4 x 4096 rays, 96 coarse samples -> 95 mid-points, 94 weights, 96 draws; peaked weights like a surface
    g = torch.Generator().manual_seed(12)
    rays, steps = 4 * 4096, 96
    z = torch.sort(torch.rand(rays, steps, generator=g) * 1.05 + 2.25, dim=1)[0]
    w = torch.exp(-((torch.arange(steps)[None] - torch.randint(5, 90, (rays, 1), generator=g)) / 3.0) ** 2) * torch.rand(rays, steps, generator=g)
    w[::7, :40] = 0
    bins, wts = 0.5 * (z[:, :-1] + z[:, 1:]), (w + 1e-5)[:, 1:-1]
    u = torch.rand(rays, steps, generator=g)
    got = vr.sample_pdf(bins.to(gpu_device), wts.to(gpu_device), steps, u=u.to(gpu_device))
    want = oracle_ops.sample_pdf(bins, wts, u)
    # `denom < eps -> 1` and the bin search make the reference function discontinuous: where a bin's mass is within an ulp
    # of eps, two correct float evaluations can land on different branches.  Element-wise agreement is therefore required
    # for all but a 1e-4 fraction, and every sample must satisfy the defining property  cdf(sample) = u  (checked against
    # a float64 piece-wise linear cdf; on either branch the deviation is bounded by the mass eps of the bin).
    err = (got.cpu() - want).abs()
    assert float((err > 5e-6).float().mean()) < 1e-4, f'sample_pdf at the benchmark shape: {int((err > 5e-6).sum())} elements differ'
    def cdf_at(smp):
        pd = (wts.double() + 1e-5)
        cd = torch.cat([torch.zeros(rays, 1, dtype=torch.float64), torch.cumsum(pd / pd.sum(-1, keepdim=True), -1)], -1)
        bd = bins.double()
        i = (torch.searchsorted(bd, smp.double().contiguous(), right=True) - 1).clamp(0, bd.shape[1] - 2)
        b0, b1 = torch.gather(bd, 1, i), torch.gather(bd, 1, i + 1)
        c0, c1 = torch.gather(cd, 1, i), torch.gather(cd, 1, i + 1)
        # tolerance: the mass eps of a bin on the `denom -> 1` branch + a float32 ulp of the depth (2.4e-7 near 3) times the
        # steepest cdf slope of the ray
        steepest = ((cd[:, 1:] - cd[:, :-1]) / (bd[:, 1:] - bd[:, :-1])).max(-1, keepdim=True)[0]
        return c0 + (smp.double() - b0) / (b1 - b0) * (c1 - c0), 2e-5 + steepest * 5e-7
    f_got, f_tol = cdf_at(got.cpu())
    assert bool(((f_got - u.double()).abs() <= f_tol).all()), 'cdf(sample) != u'
    assert bool((got.cpu() >= bins[:, :1] - 1e-6).all() and (got.cpu() <= bins[:, -1:] + 1e-6).all()), 'samples stay inside the bins'
    # size-independent property: sorted draws give sorted samples (the inverse cdf is monotone)
    us = torch.sort(u, dim=1)[0]
    srt = vr.sample_pdf(bins.to(gpu_device), wts.to(gpu_device), steps, u=us.to(gpu_device)).cpu()
    assert bool((srt[:, 1:] >= srt[:, :-1] - 1e-6).all())
    # degenerate shapes: one bin, one draw, shared draws, more weights than one wave pass, no rays
    for k, n_imp, r in ((1, 1, 3), (1, 5, 2), (200, 7, 5), (2048, 64, 2)):
        b = torch.sort(torch.rand(r, k + 1, generator=g), dim=1)[0]
        ww = torch.rand(r, k, generator=g)
        uu = torch.rand(n_imp, generator=g)
        got = vr.sample_pdf(b.to(gpu_device), ww.to(gpu_device), n_imp, u=uu.to(gpu_device))
        assert_close(got, oracle_ops.sample_pdf(b, ww, uu[None]), rtol=0, atol=2e-6, what=f'sample_pdf k={k} n={n_imp}')
    assert vr.sample_pdf(torch.zeros(0, 5, device=gpu_device), torch.zeros(0, 4, device=gpu_device), 3, det=True).shape == (0, 3)
    with pytest.raises(RuntimeError):
        hip_plugin.VolumeRenderPlugin.sample_pdf(torch.zeros(2, 4000, device=gpu_device), torch.zeros(2, 3999, device=gpu_device),
                                                 torch.zeros(4, device=gpu_device))


@pytest.mark.parametrize('n,cin,cout,h,w,k', [(2, 8, 16, 4, 4, 3), (1, 20, 150, 19, 33, 3), (3, 32, 96, 16, 16, 1), (2, 64, 3, 40, 24, 1),
                                            (1, 128, 128, 32, 32, 3), (2, 6, 19, 9, 70, 1),
                                            # flattened 1x1 tiles with a ragged last tile and rows that are not multiples of 4;
                                            # multi-image tiles (2x8x8, 8x4x4) with a ragged image group: partial 16-byte epilogue stores
                                            (2, 24, 40, 10, 14, 1), (5, 16, 48, 6, 6, 3), (9, 8, 40, 3, 5, 3), (1, 4, 200, 7, 129, 3)])
def test_modconv2d_against_conv2d(gpu_device, n, cin, cout, h, w, k):
    from torch_utils import hip_plugin
    g = torch.Generator().manual_seed(12)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g)
    s = torch.randn(n, cin, generator=g) + 1
    noise = torch.randn(h, w, generator=g)
    bias = torch.randn(cout, generator=g)
    wmod = wt[None] * s[:, None, :, None, None]
    d = (wmod.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
    ref = torch.cat([torch.nn.functional.conv2d(x[i:i + 1].double(), (wmod[i] * d[i][:, None, None, None]).double(), padding=k // 2) for i in range(n)])
    ref = ref + 0.3 * noise.double() + bias.double()[None, :, None, None]
    ref = torch.where(ref > 0, ref, ref * 0.2) * math.sqrt(2)
    ref = ref.clamp(-3.0, 3.0)
    y = hip_plugin.ModconvPlugin.modconv2d(x.to(gpu_device), wt.to(gpu_device), s.to(gpu_device), d.to(gpu_device), noise.to(gpu_device), 0.3,
                                           bias.to(gpu_device), 3, 0.2, math.sqrt(2), 3.0)
    scale = float(ref.abs().max())
    assert_close(y, ref.float(), rtol=1e-4, atol=2e-5 * max(scale, 1.0), what=f'modconv {n, cin, cout, h, w, k}')


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 8, 16, 4, 4), (1, 20, 150, 9, 13), (4, 512, 512, 8, 8), (2, 64, 32, 33, 20), (3, 12, 130, 5, 7),
                                            (4, 512, 512, 4, 4), (2, 16, 64, 3, 3), (2, 40, 96, 4, 6)])
def test_modconv2d_transposed_against_conv_transpose2d(gpu_device, n, cin, cout, h, w):
    """mode 2 == conv_transpose2d(x * s, w.transpose(0, 1), stride=2) * d  (conv2d_resample.py:114-125); also covers
    split-K (512 channels at 8x8 and 4x4: the all-class form starts at 4x4 maps), the per-class form (< 64-row blocks, maps under
    4 pixels) and the 2x8x8 / 8x4x4 pixel tiles."""
    from torch_utils import hip_plugin
    g = torch.Generator().manual_seed(14)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g)
    s = torch.randn(n, cin, generator=g) + 1
    d = torch.rand(n, cout, generator=g) + 0.5
    ref = torch.nn.functional.conv_transpose2d((x * s[:, :, None, None]).double(), wt.transpose(0, 1).double(), stride=2) * d.double()[:, :, None, None]
    y = hip_plugin.ModconvPlugin.modconv2d(x.to(gpu_device), wt.to(gpu_device), s.to(gpu_device), d.to(gpu_device), None, 0.0, None,
                                           1, 0.0, 1.0, -1.0, mode=2)
    assert y.shape == (n, cout, 2 * h + 1, 2 * w + 1)
    scale = float(ref.abs().max())
    assert_close(y, ref.float(), rtol=1e-4, atol=2e-5 * max(scale, 1.0), what=f'tconv {n, cin, cout, h, w}')
    # weights cached in packed form: a second call (weights_packed = 1) gives the same answer
    y2 = hip_plugin.ModconvPlugin.modconv2d(x.to(gpu_device), wt.to(gpu_device), s.to(gpu_device), d.to(gpu_device), None, 0.0, None,
                                            1, 0.0, 1.0, -1.0, mode=2)
    assert torch.equal(y, y2)


@pytest.mark.parametrize('n,cin,cout,h,w', [(4, 128, 128, 128, 128), (3, 64, 64, 32, 32), (2, 16, 24, 9, 6)])
def test_modconv2d_transposed_per_image_weights(gpu_device, n, cin, cout, h, w):
    """mode 2 with PER-IMAGE weights [n, cout, cin, 3, 3] (styles folded in by the caller): every image is convolved with ITS weights in
    the whole output, including output row 2h and column 2w (the strip plan — `tconv_strip_kernel` reads one shared weight tensor — must not
    be chosen for such a launch: ADVICE r4; `modconv_plan` agrees)."""
    from torch_utils import hip_plugin
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(n, cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    d = torch.rand(n, cout, generator=g) + 0.5
    ref = torch.stack([torch.nn.functional.conv_transpose2d(x[i:i + 1].double(), wt[i].transpose(0, 1).double(), stride=2)[0] for i in range(n)])
    ref = ref * d.double()[:, :, None, None]
    plan = hip_plugin.modconv_plan(n, cin, cout, h, w, mode=2, per_image=True, epilogue='plain')
    assert not plan['strip'], plan
    y = hip_plugin.ModconvPlugin.modconv2d(x.to(gpu_device), wt.to(gpu_device), None, d.to(gpu_device), None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=2)
    assert y.shape == (n, cout, 2 * h + 1, 2 * w + 1)
    scale = float(ref.abs().max())
    for name, sl in (('interior', (slice(None), slice(None), slice(0, 2 * h), slice(0, 2 * w))), ('last row', (slice(None), slice(None), slice(2 * h, None))),
                     ('last column', (slice(None), slice(None), slice(None), slice(2 * w, None)))):
        assert_close(y[sl], ref[sl].float(), rtol=1e-4, atol=2e-5 * max(scale, 1.0), what=f'per-image tconv {n, cin, cout, h, w}: {name}')


def test_modconv2d_low_resolution_split_k(gpu_device):
    """512 -> 512 channels at 4x4 / 8x8 / 16x16 / 32x32, batch 4: split-K path + image-batched pixel tiles."""
    from torch_utils import hip_plugin
    g = torch.Generator().manual_seed(15)
    for res in (4, 8, 16, 32):
        x = torch.randn(4, 512, res, res, generator=g); wt = torch.randn(512, 512, 3, 3, generator=g) * 0.05
        s = torch.randn(4, 512, generator=g) + 1; d = torch.rand(4, 512, generator=g) + 0.5
        nz = torch.randn(res, res, generator=g); b = torch.randn(512, generator=g)
        ref = torch.nn.functional.conv2d((x * s[:, :, None, None]).double(), wt.double(), padding=1) * d.double()[:, :, None, None]
        ref = ref + 0.5 * nz.double() + b.double()[None, :, None, None]
        ref = torch.where(ref > 0, ref, ref * 0.2) * math.sqrt(2)
        y = hip_plugin.ModconvPlugin.modconv2d(x.to(gpu_device), wt.to(gpu_device), s.to(gpu_device), d.to(gpu_device), nz.to(gpu_device), 0.5,
                                               b.to(gpu_device), 3, 0.2, math.sqrt(2), -1.0)
        assert_close(y, ref.float(), rtol=1e-4, atol=2e-5 * float(ref.abs().max()), what=f'res {res}')


def test_modulated_conv2d_surface(golden, gpu_device):
    from training import networks
    from torch_utils.ops import upfirdn2d
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).to(gpu_device)
    with torch.no_grad():
        for cfg, a in golden('networks').select(fn='modulated_conv2d'):
            kw = {k: cfg[k] for k in ('up', 'padding', 'demodulate', 'fused_modconv', 'flip_weight')}
            y = networks.modulated_conv2d(x=t(a['in_x'], gpu_device), weight=t(a['in_w'], gpu_device), styles=t(a['in_s'], gpu_device),
                                          noise=t(a['in_noise'], gpu_device), resample_filter=(f if cfg['up'] > 1 else None), **kw)
            assert_close(y, a['out_y'], rtol=1e-4, atol=1e-4, what=str(cfg))
    assert _calls('modconv2d') >= 3


# ---- frame conversion ------------------------------------------------------------------------------------------------------

def test_frame_u8(golden, gpu_device):
    from torch_utils import hip_plugin
    (cfg, a), = golden('post').select(fn='frame')
    pal = torch.from_numpy(oracle_ops.PALETTE).to(gpu_device)
    out = hip_plugin.FramePlugin.frame_u8(t(a['in_img'], gpu_device), t(a['in_seg'], gpu_device), pal).cpu().numpy()
    grid = np.concatenate([out[0], out[1]], axis=1)
    assert np.array_equal(grid, a['out_grid_u8'])
    g = torch.Generator().manual_seed(13)
    img = torch.randn(3, 3, 64, 128, generator=g); seg = torch.randn(3, 19, 64, 128, generator=g)
    out = hip_plugin.FramePlugin.frame_u8(img.to(gpu_device), seg.to(gpu_device), pal).cpu().numpy()
    assert np.array_equal(out, oracle_ops.frame_u8(img, seg))


# ---- camera poses (csrc/camera.hip) ------------------------------------------------------------------------------------

def test_camera_pose_kernels(golden, gpu_device, monkeypatch):
    """`sample_camera_positions`, `create_cam2world_matrix`, `LookAtPoseSampler.sample` on device tensors = two launches (csrc/camera.hip):
    the reference-run fixtures (volumetric_rendering.py:147-213, 268-295, run on the CPU) and the same functions spelled as tensor operations
    (`IDE3D_NO_CAMERA_KERNELS=1`), every sampling mode under one generator state, one camera to several workgroups of them."""
    import math
    from training import volumetric_rendering as vr
    from torch_utils import hip_plugin
    dev = gpu_device
    calls = lambda: (hip_plugin.CALLS.get('sphere_points', 0), hip_plugin.CALLS.get('cam2world', 0))
    for cfg, a in golden('volumetric'):
        before = calls()
        if cfg['fn'] == 'gen_images_pose':
            cam, phi, theta = vr.sample_camera_positions(dev, n=1, r=2.7, horizontal_mean=cfg['yaw'] + math.pi / 2, vertical_mean=math.pi / 2, mode=None)
            assert_close(vr.create_cam2world_matrix(-cam, cam, device=dev), a['out_c2w'], rtol=0, atol=1e-6, what='c2w')
            assert calls() == (before[0] + 1, before[1] + 1) and phi.shape == theta.shape == (1, 1)
        elif cfg['fn'] == 'lookat':
            c2w = vr.LookAtPoseSampler.sample(cfg['h'], cfg['v'], torch.tensor(cfg['lookat'], device=dev), radius=cfg['radius'], device=dev)
            assert_close(c2w, a['out_c2w'], rtol=0, atol=2e-6, what='lookat')
            assert calls() == (before[0] + 1, before[1] + 1)

    def both(fn):
        torch.manual_seed(5); random_state = __import__('random').getstate()
        monkeypatch.delenv('IDE3D_NO_CAMERA_KERNELS', raising=False)
        got = fn()
        torch.manual_seed(5); __import__('random').setstate(random_state)
        monkeypatch.setenv('IDE3D_NO_CAMERA_KERNELS', '1')
        before = calls()
        ref = fn()
        assert calls() == before                                              # the switch really selects the tensor operations
        monkeypatch.delenv('IDE3D_NO_CAMERA_KERNELS')
        for g, r in zip(got if isinstance(got, tuple) else (got,), ref if isinstance(ref, tuple) else (ref,)):
            assert g.shape == r.shape and g.dtype == r.dtype and g.device == r.device
            assert_close(g, r, rtol=0, atol=3e-6, what=fn.__name__ if hasattr(fn, '__name__') else 'pose')

    for n in (1, 5, 200):
        for mode in (None, 'uniform', 'normal', 'hybrid', 'truncated_gaussian', 'spherical_uniform'):
            both(lambda: vr.sample_camera_positions(dev, n=n, r=1.7, horizontal_stddev=0.4, vertical_stddev=0.2, horizontal_mean=1.1, vertical_mean=1.9, mode=mode))
        g = torch.Generator().manual_seed(n)
        fwd = torch.randn(n, 3, generator=g).to(dev); org = torch.randn(n, 3, generator=g).to(dev)
        both(lambda: vr.create_cam2world_matrix(fwd, org, device=dev))
        for lookat in (torch.tensor([0.0, 0.05, 0.2], device=dev), torch.tensor([[0.1, 0.0, -0.2]], device=dev), torch.randn(n, 3, generator=g).to(dev) * 0.1):
            both(lambda: vr.LookAtPoseSampler.sample(1.3, 1.4, lookat, horizontal_stddev=0.3, vertical_stddev=0.2, radius=2.7, batch_size=n, device=dev))
    # pitch outside (0, pi) is clamped like torch.clamp; a pose that takes part in autograd keeps the tensor operations
    both(lambda: vr.sample_camera_positions(dev, n=3, r=1.0, horizontal_mean=0.3, vertical_mean=-4.0, mode=None))
    both(lambda: vr.LookAtPoseSampler.sample(0.2, 7.0, torch.zeros(3, device=dev), radius=1.0, batch_size=2, device=dev))
    before = calls()
    fwd = torch.randn(2, 3, device=dev, requires_grad=True)
    vr.create_cam2world_matrix(fwd, torch.zeros(2, 3, device=dev), device=dev).sum().backward()
    assert calls() == before and fwd.grad is not None


# ---- style preparation (affine styles, demodulation coefficients, folded head weights) ---------------------------------

def test_style_demod_and_fold_heads_vs_formula(gpu_device):
    """`ide3d_style_demod` / `ide3d_fold_heads` == the framework formulas of networks.py:52-60,91-93,700-706 (fp64 reference),
    over batch sizes that cross the 8-images-per-launch grouping, channel counts that are not multiples of the block sizes,
    strided ws rows and the tiny w_dim."""
    from torch_utils import hip_plugin
    g = torch.Generator().manual_seed(31)
    for n, cin, cout, wdim in ((4, 512, 512, 512), (9, 48, 40, 32), (1, 128, 64, 512), (3, 64, 22, 36)):
        ws = torch.randn(n, 3, wdim, generator=g).to(gpu_device)
        w = ws[:, 1]                                                    # strided rows, unit inner stride
        A = torch.randn(cin, wdim, generator=g).to(gpu_device)
        b = torch.randn(cin, generator=g).to(gpu_device)
        W = torch.randn(cout, cin, 3, 3, generator=g).to(gpu_device)
        wsq_t = W.double().square().sum(dim=[2, 3]).t().contiguous().float()
        a_gain, b_gain = 1 / math.sqrt(wdim), 1.0
        styles, dcoefs = hip_plugin.StylePlugin.style_demod(w, A, b, a_gain, b_gain, wsq_t)
        s_ref = w.double() @ (A.double() * a_gain).t() + b.double() * b_gain
        d_ref = ((W.double().unsqueeze(0) * s_ref.reshape(n, 1, cin, 1, 1)).square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
        assert_close(styles, s_ref.float().cpu(), rtol=2e-5, atol=2e-5, what=f'styles {n, cin, cout, wdim}')
        assert_close(dcoefs, d_ref.float().cpu(), rtol=1e-4, atol=1e-7, what=f'dcoefs {n, cin, cout, wdim}')
        only, none = hip_plugin.StylePlugin.style_demod(w, A, b, a_gain, b_gain, None)
        assert none is None and torch.equal(only, styles)
        # two heads (rgb 3, seg 19) folded into per-image 1x1 weights
        A1 = torch.randn(cin, wdim, generator=g).to(gpu_device); b1 = torch.randn(cin, generator=g).to(gpu_device)
        W0 = torch.randn(3, cin, generator=g).to(gpu_device); W1 = torch.randn(19, cin, generator=g).to(gpu_device)
        g0, g1 = 1 / math.sqrt(cin), 0.5 / math.sqrt(cin)
        out = hip_plugin.StylePlugin.fold_heads(w, a_gain, A, b, W0, g0, A1, b1, W1, g1)
        s0 = (w.double() @ (A.double() * a_gain).t() + b.double()) * g0
        s1 = (w.double() @ (A1.double() * a_gain).t() + b1.double()) * g1
        ref = torch.cat([W0.double().unsqueeze(0) * s0.unsqueeze(1), W1.double().unsqueeze(0) * s1.unsqueeze(1)], dim=1)
        assert out.shape == (n, 22, cin, 1, 1)
        assert_close(out.reshape(n, 22, cin), ref.float().cpu(), rtol=2e-5, atol=2e-5, what=f'fold_heads {n, cin, wdim}')


def test_upfirdn2d_fused_epilogue(gpu_device):
    """`ide3d_upfirdn2d_ex`: FIR + (skip add | noise * strength | bias + lrelu * gain + clamp) in the store of the FIR ==
    the separate reference ops (upfirdn2d.py:167 + networks.py:404-414 / 1100-1121), on shapes that hit aligned 16-byte
    rows, ragged row ends, odd widths (scalar epilogue), the 64-wide tile, and the generic kernel (5-tap filter)."""
    from torch_utils import hip_plugin
    from torch_utils.ops import upfirdn2d as up, bias_act
    g = torch.Generator().manual_seed(41)
    P = hip_plugin.Upfirdn2dPlugin
    f4 = up.setup_filter([1, 3, 3, 1], device=gpu_device)
    f5 = up.setup_filter([1, 4, 6, 4, 1], device=gpu_device)
    # post-transposed-conv FIR (up 1, pad 1, gain 4) + noise + bias + lrelu + clamp
    for shape, f in (((2, 5, 65, 65), f4), ((1, 3, 131, 131), f4), ((2, 4, 38, 43), f4), ((1, 2, 21, 30), f5)):
        x = torch.randn(*shape, generator=g).to(gpu_device)
        pad = 1 if f is f4 else 2
        oh, ow = shape[2] + 2 * pad - f.shape[0] + 1, shape[3] + 2 * pad - f.shape[0] + 1
        noise = torch.randn(oh, ow, generator=g).to(gpu_device)
        b = torch.randn(shape[1], generator=g).to(gpu_device)
        got = P.upfirdn2d_ex(x, f, 1, 1, 1, 1, pad, pad, pad, pad, False, 4.0, noise=noise, noise_strength=0.37, bias=b, act=3, alpha=0.2,
                             act_gain=math.sqrt(2), clamp=1.5)
        ref = up._upfirdn2d_ref(x.cpu().double(), f.cpu(), padding=[pad] * 4, gain=4)
        ref = bias_act._bias_act_ref(ref + noise.cpu().double() * 0.37, b.cpu().double(), act='lrelu', gain=math.sqrt(2), clamp=1.5)
        assert_close(got, ref.float(), rtol=1e-5, atol=1e-5, what=f'fir+noise+bias_act {shape}')
    # skip-image upsample (up 2, pad [2,1,2,1], gain 4) + add, with a non-contiguous addend
    for shape in ((2, 22, 16, 16), (1, 3, 33, 20), (3, 96, 8, 8)):
        x = torch.randn(*shape, generator=g).to(gpu_device)
        big = torch.randn(shape[0], shape[1], 2 * shape[2], 2 * shape[3] + 3, generator=g).to(gpu_device)
        for add in (big[..., :2 * shape[3]].contiguous(), big[..., 1:2 * shape[3] + 1]):
            got = P.upfirdn2d_ex(x, f4, 2, 2, 1, 1, 2, 1, 2, 1, False, 4.0, add=add)
            ref = up._upfirdn2d_ref(x.cpu().double(), f4.cpu(), up=2, padding=[2, 1, 2, 1], gain=4) + add.cpu().double()
            assert_close(got, ref.float(), rtol=1e-5, atol=1e-5, what=f'up2+add {shape}')


def test_lean_fir_kernels_equal_the_tile_kernels_bit_for_bit(gpu_device):
    """Round 5: `fir44_kernel` (fp32 4x4 FIR at unit rate behind every transposed convolution: padded 16-byte aligned rows, out_w % 4 == 0)
    and `fir_up2_kernel` (the skip upsampler) compute the tile kernels' values in the tile kernels' order with the bookkeeping taken out
    of the vector pipe.  IDE3D_FIR_NO_LEAN=1 (read per call) routes the same launch to the tile kernel: outputs and `y_amax` must be
    bit-equal for every epilogue, both flips, ragged tiles on both edges; and within 1e-5 of the float64 definition."""
    import os
    from torch_utils import hip_plugin
    from torch_utils.ops import upfirdn2d as up, bias_act
    P = hip_plugin.Upfirdn2dPlugin
    g = torch.Generator().manual_seed(77)
    rn = lambda *sh: torch.randn(*sh, generator=g).to(gpu_device)
    f4 = up.setup_filter([1, 3, 3, 1], device=gpu_device)
    fa = (torch.tensor([[1., 2., 3., 4.], [0.5, 3., 3., 1.], [2., 3., 5., 1.], [1., 3., 3., 7.]]) / 40).to(gpu_device)     # asymmetric: catches a flipped tap

    def both(fn):
        try:
            os.environ['IDE3D_FIR_NO_LEAN'] = '1'
            ref = fn()
        finally:
            os.environ.pop('IDE3D_FIR_NO_LEAN', None)
        return fn(), ref

    checked = 0
    for n, c, h, w in ((2, 6, 513, 513), (1, 5, 77, 201), (3, 7, 129, 129), (1, 2, 257, 133)):
        x = torch.nn.functional.pad(rn(n, c, h, w), (0, (-w) % 4))[..., :w]              # rows padded to 16 bytes, like the transposed convolution writes them
        nz, bb = rn(h - 1, w - 1), rn(c)
        for f in (f4, fa):
            for flip in (False, True):
                for kw in ({}, dict(noise=nz, noise_strength=0.7, bias=bb, act=3, alpha=0.2, act_gain=math.sqrt(2), clamp=-1.0),
                           dict(noise=nz, noise_strength=0.7, bias=bb, act=3, alpha=0.2, act_gain=math.sqrt(2), clamp=0.8),
                           dict(bias=bb, act=1, alpha=0.0, act_gain=1.0, clamp=-1.0), dict(bias=bb, act=3, alpha=1.5, act_gain=1.0, clamp=-1.0)):
                    def run():
                        am = torch.zeros(n, hip_plugin.AMAX_FLOATS, device=gpu_device)
                        return P.upfirdn2d_ex(x, f, 1, 1, 1, 1, 1, 1, 1, 1, flip, 4.0, y_amax=am, **kw), am.amax(dim=1)
                    (y, am), (yr, amr) = both(run)
                    assert torch.equal(y, yr) and torch.equal(am, amr), f'fir44 {(n, c, h, w)} flip {flip} {sorted(kw)}'
                    assert torch.equal(am, y.abs().amax(dim=(1, 2, 3)))
                    checked += 1
        ref = up._upfirdn2d_ref(x.cpu().double(), f4.cpu(), padding=[1] * 4, gain=4)
        ref = bias_act._bias_act_ref(ref + nz.cpu().double() * 0.7, bb.cpu().double(), act='lrelu', gain=math.sqrt(2))
        got = P.upfirdn2d_ex(x, f4, 1, 1, 1, 1, 1, 1, 1, 1, False, 4.0, noise=nz, noise_strength=0.7, bias=bb, act=3, alpha=0.2, act_gain=math.sqrt(2), clamp=-1.0)
        assert_close(got, ref.float(), rtol=1e-5, atol=1e-5, what=f'fir44 + epilogue {(n, c, h, w)}')
    for n, c, h, w in ((2, 12, 128, 128), (1, 22, 256, 256), (3, 5, 8, 8), (2, 3, 20, 36), (1, 4, 64, 200)):
        x, add, bb = rn(n, c, h, w), rn(n, c, 2 * h, 2 * w), rn(c)
        for f in (f4, fa):
            for flip in (False, True):
                for kw in ({}, dict(add=add), dict(add=add, bias=bb, act=3, alpha=0.2, act_gain=1.3, clamp=0.9)):
                    (y, yr) = both(lambda: P.upfirdn2d_ex(x, f, 2, 2, 1, 1, 2, 1, 2, 1, flip, 4.0, **kw))
                    assert torch.equal(y, yr), f'fir_up2 {(n, c, h, w)} flip {flip} {sorted(kw)}'
                    checked += 1
        ref = up._upfirdn2d_ref(x.cpu().double(), f4.cpu(), up=2, padding=[2, 1, 2, 1], gain=4) + add.cpu().double()
        assert_close(P.upfirdn2d_ex(x, f4, 2, 2, 1, 1, 2, 1, 2, 1, False, 4.0, add=add), ref.float(), rtol=1e-5, atol=1e-5, what=f'fir_up2 + add {(n, c, h, w)}')
    assert checked == 4 * 2 * 2 * 5 + 5 * 2 * 2 * 3


def test_modconv2d_packed_weight_cache_is_identity_safe(gpu_device):
    """The packed-weight workspace is reused only for the same tensor object at the same version: an in-place update, or a
    different weight that lands on a recycled address, must be re-packed."""
    from torch_utils import hip_plugin
    g = torch.Generator().manual_seed(51)
    x = torch.randn(1, 16, 16, 16, generator=g).to(gpu_device)
    s = torch.ones(1, 16, device=gpu_device)

    def run(w):
        return hip_plugin.ModconvPlugin.modconv2d(x, w, s, None, None, 0.0, None, 1, 0.0, 1.0, -1.0)

    def ref(w):
        return torch.nn.functional.conv2d(x, w, padding=1)

    w1 = torch.randn(32, 16, 3, 3, generator=g).to(gpu_device)
    assert_close(run(w1), ref(w1).cpu(), rtol=1e-4, atol=1e-4, what='first weight')
    assert_close(run(w1), ref(w1).cpu(), rtol=1e-4, atol=1e-4, what='cached pack')
    w1.mul_(2.0)                                                   # in-place update bumps _version
    assert_close(run(w1), ref(w1).cpu(), rtol=1e-4, atol=1e-4, what='after in-place update')
    ptr = w1.data_ptr()
    del w1
    for _ in range(8):                                             # the caching allocator hands the block out again
        w2 = torch.randn(32, 16, 3, 3, generator=g).to(gpu_device)
        if w2.data_ptr() == ptr:
            break
    assert_close(run(w2), ref(w2).cpu(), rtol=1e-4, atol=1e-4, what='new weight (possibly at the recycled address)')


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 16, 32, 19, 23), (1, 64, 128, 65, 65), (4, 8, 40, 7, 9), (3, 32, 64, 4, 3), (1, 512, 512, 17, 17)])
def test_modconv2d_stride2_against_conv2d(gpu_device, n, cin, cout, h, w):
    """mode 1: 3x3 stride-2 convolution without padding + bias + lrelu == F.conv2d (fp64), plain (styles / dcoefs NULL) and modulated."""
    from torch_utils import hip_plugin
    g = torch.Generator().manual_seed(61)
    x = torch.randn(n, cin, h, w, generator=g).to(gpu_device)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(gpu_device)
    b = torch.randn(cout, generator=g).to(gpu_device)
    got = hip_plugin.ModconvPlugin.modconv2d(x, wt, None, None, None, 0.0, b, 3, 0.2, math.sqrt(2), -1.0, mode=1)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x.double().cpu(), wt.double().cpu(), b.double().cpu(), stride=2), 0.2) * math.sqrt(2)
    assert got.shape == ref.shape
    assert_close(got, ref.float(), rtol=2e-4, atol=2e-4, what='stride-2 conv')
    s = (torch.randn(n, cin, generator=g) + 1).to(gpu_device); d = torch.rand(n, cout, generator=g).to(gpu_device)
    got = hip_plugin.ModconvPlugin.modconv2d(x, wt, s, d, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=1)
    ref = torch.nn.functional.conv2d((x.double() * s.double()[:, :, None, None]).cpu(), wt.double().cpu(), stride=2) * d.double().cpu()[:, :, None, None]
    assert_close(got, ref.float(), rtol=2e-4, atol=2e-4, what='modulated stride-2 conv')


# ---- gradfix surfaces (custom autograd graphs; only active on device tensors with the module switch on) -----------------

def test_conv2d_gradfix_second_order(gpu_device):
    """conv2d_gradfix with `enabled`: outputs, first- and second-order gradients (R1-style: grad of |dy/dx|^2 w.r.t. the
    weight) equal plain autograd through torch.nn.functional; `no_weight_gradients` suppresses the weight gradient."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(71)
    cases = [(False, dict(stride=1, padding=1)), (False, dict(stride=2, padding=1)), (True, dict(stride=2, padding=0)),
             (False, dict(stride=1, padding=0, groups=2))]
    for transpose, kw in cases:
        groups = kw.get('groups', 1)
        x0 = torch.randn(2, 4, 9, 10, generator=g).double().to(gpu_device)
        w0 = (torch.randn(4, 4 // groups, 3, 3, generator=g) if not transpose else torch.randn(4, 6, 3, 3, generator=g)).double().to(gpu_device)
        res = []
        for use_custom in (True, False):
            cg.enabled = use_custom
            try:
                x = x0.clone().requires_grad_(True); w = w0.clone().requires_grad_(True)
                fn = cg.conv_transpose2d if transpose else cg.conv2d
                y = fn(x, w, **kw)
                (gx,) = torch.autograd.grad(y.square().sum(), [x], create_graph=True)
                pen = gx.square().sum()
                gw, gx2 = torch.autograd.grad(pen, [w, x])
                res.append((y.detach(), gx.detach(), gw, gx2))
            finally:
                cg.enabled = False
        for a, b, name in zip(res[0], res[1], ('y', 'dx', 'd(pen)/dw', 'd(pen)/dx')):
            assert_close(a, b.cpu(), rtol=1e-9, atol=1e-9, what=f'{name} transpose={transpose} {kw}')
    cg.enabled = True
    try:
        x = x0.clone().requires_grad_(True); w = w0.clone().requires_grad_(True)
        with cg.no_weight_gradients():
            y = cg.conv_transpose2d(x, w, stride=2) if transpose else cg.conv2d(x, w, **kw)
            gx, gw = torch.autograd.grad(y.sum(), [x, w], allow_unused=True)
        assert gw is None and gx is not None and not cg.weight_gradients_disabled
    finally:
        cg.enabled = False


def test_grid_sample_gradfix_second_order(gpu_device):
    """grid_sample_gradfix with `enabled`: value and first-order gradients equal plain autograd through F.grid_sample;
    the second-order path (which plain autograd does not have) satisfies the analytic identity
    d/dp |A^T p|^2 = 2 A A^T p for the linear look-up A = sample(., grid)."""
    from torch_utils.ops import grid_sample_gradfix as gs
    g = torch.Generator().manual_seed(72)
    img0 = torch.randn(2, 3, 7, 9, generator=g).double().to(gpu_device)
    grid0 = (torch.rand(2, 5, 6, 2, generator=g) * 2.2 - 1.1).double().to(gpu_device)
    wgt = torch.randn(2, 3, 5, 6, generator=g).double().to(gpu_device)
    res = []
    for use_custom in (True, False):
        gs.enabled = use_custom
        try:
            img = img0.clone().requires_grad_(True); grid = grid0.clone().requires_grad_(True)
            y = gs.grid_sample(img, grid)
            d_img, d_grid = torch.autograd.grad((y * wgt).sum(), [img, grid])
            res.append((y.detach(), d_img, d_grid))
        finally:
            gs.enabled = False
    for a_, b_, name in zip(res[0], res[1], ('y', 'd image', 'd grid')):
        assert_close(a_, b_.cpu(), rtol=1e-9, atol=1e-9, what=name)
    gs.enabled = True
    try:
        img = img0.clone().requires_grad_(True)
        probe = wgt.clone().requires_grad_(True)
        y = gs.grid_sample(img, grid0)
        (d_img,) = torch.autograd.grad((y * probe).sum(), [img], create_graph=True)       # A^T p
        (d_probe,) = torch.autograd.grad(d_img.square().sum(), [probe])
    finally:
        gs.enabled = False
    expect = 2 * torch.nn.functional.grid_sample(d_img.detach(), grid0, mode='bilinear', padding_mode='zeros', align_corners=False)
    assert_close(d_probe, expect.cpu(), rtol=1e-9, atol=1e-9, what='second order')


def test_triplane_image_groups_beyond_2gib(gpu_device):
    """Planes larger than 2 GiB in total: both gather kernels address images in groups whose byte offsets fit 32 bits
    (6 images x 403 MB -> groups of 5 + 1); flat == tiled, and a subset equals the oracle (last image included)."""
    from dnnlib import util
    n, C, H = 6, 32, 1024
    g = torch.Generator(device=gpu_device).manual_seed(81)
    planes = torch.randn(n, 3 * C, H, H, generator=g, device=gpu_device).contiguous(memory_format=torch.channels_last)
    assert planes.numel() * 4 > 2 ** 31
    co = (torch.rand(n, 16 * 16 * 8, 3, generator=g, device=gpu_device) * 2 - 1) * 0.9
    flat = util.sample_from_triplane(co, planes)
    tiled = util.sample_from_triplane(co, planes, ray_grid=(16, 16, 8))
    assert torch.equal(flat, tiled)
    idx = torch.arange(0, co.shape[1], 37, device=gpu_device)
    for img in (0, 4, 5):
        ref = fast_ops.sample_from_triplane(co[img:img + 1, idx].cpu(), planes[img:img + 1].cpu().contiguous())
        got = flat.reshape(n, -1, C)[img, idx]
        assert_close(got, ref, rtol=1e-5, atol=2e-5, what=f'image {img}')


# ---- resampling between the big kernels (csrc/resample.hip) -----------------------------------------------------------

@pytest.mark.parametrize('n,c,h,w', [(2, 96, 32, 32), (1, 36, 13, 10), (3, 8, 5, 33)])
def test_skip_upsample_add_channels_last(gpu_device, n, c, h, w):
    """upsample2d(lo, [1,3,3,1]) + add written channels-last == the oracle's upsample2d + add (and == the NCHW HIP FIR path);
    `add` is a channel slice of a wider tensor like the dual-head output; partial tiles and partial channel groups included."""
    from torch_utils import hip_plugin
    from torch_utils.ops import upfirdn2d
    g = torch.Generator().manual_seed(21)
    lo = torch.randn(n, c, h, w, generator=g)
    wide = torch.randn(n, 2 * c, 2 * h, 2 * w, generator=g)
    add = wide[:, c:]
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    want = oracle_ops.upsample2d(lo.double(), f) + add.double()
    before = _calls('skip_upsample_add_cl')
    got = hip_plugin.ResamplePlugin.skip_upsample_add_cl(lo.to(gpu_device), wide.to(gpu_device)[:, c:])
    assert _calls('skip_upsample_add_cl') == before + 1
    assert got.shape == (n, c, 2 * h, 2 * w) and got.is_contiguous(memory_format=torch.channels_last)
    assert_close(got, want, rtol=1e-6, atol=1e-6, what='skip accumulate (channels-last)')
    nchw = upfirdn2d.upsample2d(lo.to(gpu_device), f.to(gpu_device)) + wide.to(gpu_device)[:, c:]
    assert_close(got, nchw, rtol=1e-6, atol=1e-6, what='vs NCHW HIP path')


def test_bilinear_up2_split(gpu_device):
    """One-launch 2x bilinear (align_corners=False) split into (features, raw RGB, seg logits) == F.interpolate per slice."""
    from torch_utils import hip_plugin
    g = torch.Generator().manual_seed(22)
    for n, c, h, w, ranges in ((4, 51, 64, 64, [(0, 32), (0, 3), (32, 19)]), (2, 13, 7, 10, [(0, 8), (0, 3), (8, 5)]), (1, 5, 3, 4, [(1, 4)])):
        x = torch.randn(n, c, h, w, generator=g)
        outs = hip_plugin.ResamplePlugin.bilinear_up2_split(x.to(gpu_device), ranges)
        assert len(outs) == len(ranges)
        for o, (b, cnt) in zip(outs, ranges):
            want = torch.nn.functional.interpolate(x[:, b:b + cnt], size=(2 * h, 2 * w), mode='bilinear', align_corners=False)
            assert o.shape == want.shape
            assert_close(o, want, rtol=1e-6, atol=1e-6, what=f'bilinear channels {b}+{cnt}')
        if len(ranges) == 3:          # outputs 1 and 2 as the two channel ranges of one tensor (what the super-resolution path asks for): same values
            adj = hip_plugin.ResamplePlugin.bilinear_up2_split(x.to(gpu_device), ranges, adjacent=(1, 2))
            assert adj[1]._base is adj[2]._base and adj[1]._base.shape[1] == ranges[1][1] + ranges[2][1]
            for o, a_ in zip(outs, adj):
                assert torch.equal(o, a_)


# ---- mapping network in one launch (csrc/mapping.hip) ---------------------------------------------------------------------

def test_mapping_network_single_launch(golden, gpu_device):
    """`MappingNetwork.forward` on the GPU = ONE launch of the fused kernel; equals the module's CPU definition (which is pinned
    to the reference's MappingNetwork by tests/test_host_cpu.py) for full-size and tiny widths, with and without truncation."""
    from training import networks
    for (z_dim, c_dim, w_dim, num_ws, layers) in ((512, 25, 512, 18, 8), (32, 25, 32, 12, 2), (64, 0, 48, 5, 3)):
        torch.manual_seed(3)
        M = networks.MappingNetwork(z_dim, c_dim, w_dim, num_ws, num_layers=layers).eval()
        with torch.no_grad():
            M.w_avg.copy_(torch.randn(w_dim) * 0.3)
        Mg = networks.MappingNetwork(z_dim, c_dim, w_dim, num_ws, num_layers=layers).eval()
        Mg.load_state_dict(M.state_dict()); Mg = Mg.to(gpu_device)
        g = torch.Generator().manual_seed(4)
        for n in (1, 4, 7):
            z = torch.randn(n, z_dim, generator=g); c = torch.randn(n, c_dim, generator=g) if c_dim else None
            for psi, cutoff in ((1, None), (0.7, None), (0.3, 3)):
                before = _calls('mapping')
                with torch.no_grad():
                    got = Mg(z.to(gpu_device), None if c is None else c.to(gpu_device), truncation_psi=psi, truncation_cutoff=cutoff)
                    want = M(z, c, truncation_psi=psi, truncation_cutoff=cutoff)
                assert _calls('mapping') == before + 1, 'the fused mapping kernel must have run'
                assert got.shape == (n, num_ws, w_dim)
                assert_close(got, want, rtol=1e-4, atol=1e-5, what=f'ws z{z_dim} n{n} psi{psi} cutoff{cutoff}')
    # float64 latents (what the drivers feed: np.random.RandomState(seed).randn) and batches beyond the kernel's limit
    z64 = torch.from_numpy(np.random.RandomState(0).randn(9, 64))
    with torch.no_grad():
        before = _calls('mapping')
        got = Mg(z64.to(gpu_device), None)
        assert _calls('mapping') == before, 'n = 9 > 8 takes the generic path'
        assert_close(got, M(z64, None), rtol=1e-4, atol=1e-5, what='generic path')
        assert_close(Mg(z64[:4].to(gpu_device), None), M(z64[:4], None), rtol=1e-4, atol=1e-5, what='float64 z')

def test_style_batch_equals_per_layer(gpu_device):
    """ide3d_style_demod_batch / ide3d_fold_heads_batch (all layers of a pass in 2 + 1 launches) == the per-layer entry points, bit for
    bit: mixed channel counts, a layer without demodulation, strided latents (ws[:, i] views), more jobs than one launch holds."""
    from torch_utils import hip_plugin
    g = torch.Generator().manual_seed(31)
    n, wdim = 4, 512
    ws = torch.randn(n, 30, wdim, generator=g).to(gpu_device)
    P = hip_plugin.StylePlugin
    jobs, ref = [], []
    shapes = [(512, 512), (512, 256), (256, 128), (96, 64), (40, 0), (64, 64)] * 5          # 30 jobs > STYLE_BATCH_MAX; cout 0 = no demodulation
    for k, (cin, cout) in enumerate(shapes):
        aw = (torch.randn(cin, wdim, generator=g) * 0.05).to(gpu_device)
        ab = torch.randn(cin, generator=g).to(gpu_device)
        wsq = (torch.rand(cin, cout, generator=g) + 0.1).to(gpu_device) if cout else None
        w = ws[:, k]
        jobs.append((w, aw, ab, 1 / math.sqrt(wdim), 1.0, wsq))
        ref.append(P.style_demod(w, aw, ab, 1 / math.sqrt(wdim), 1.0, wsq))
    before = hip_plugin.CALLS.get('style_demod_batch', 0)
    got = P.style_demod_batch(jobs)
    assert hip_plugin.CALLS.get('style_demod_batch', 0) - before == 2            # 24 + 6 jobs
    for (s0, d0), (s1, d1) in zip(ref, got):
        assert torch.equal(s0, s1)
        assert (d0 is None and d1 is None) or torch.equal(d0, d1)
    fjobs, fref = [], []
    for k, (cin, c0, c1) in enumerate([(512, 96, 96), (256, 96, 96), (128, 96, 96), (64, 3, 19), (128, 3, 19), (32, 3, 19)]):
        a0 = (torch.randn(cin, wdim, generator=g) * 0.05).to(gpu_device); b0 = torch.randn(cin, generator=g).to(gpu_device)
        a1 = (torch.randn(cin, wdim, generator=g) * 0.05).to(gpu_device); b1 = torch.randn(cin, generator=g).to(gpu_device)
        w0 = torch.randn(c0, cin, generator=g).to(gpu_device); w1 = torch.randn(c1, cin, generator=g).to(gpu_device)
        w = ws[:, 20 + k]
        args = (w, 1 / math.sqrt(wdim), a0, b0, w0, 1 / math.sqrt(cin), a1, b1, w1, 1 / math.sqrt(cin))
        fjobs.append(args)
        fref.append(P.fold_heads(*args))
    fgot = P.fold_heads_batch(fjobs)
    for r0, r1 in zip(fref, fgot):
        assert r0.shape == r1.shape and torch.equal(r0, r1)


def test_mapping_batch_sizes_across_the_one_block_threshold_agree(gpu_device):
    """the fused mapping kernel sums a row's products in another order when a wave owns a whole row block (small batches) than
    when it does not, so `ws` is not bitwise identical across batch sizes — only to rounding.  Rows of one latent computed at batch 1, 4, 5
    and 8 agree within a few ulp of the layer outputs' scale, and every one of them is what the layer-by-layer definition gives on the CPU."""
    from training import networks
    torch.manual_seed(5)
    M = networks.MappingNetwork(512, 25, 512, 18).eval()
    z = torch.randn(8, 512); c = torch.randn(8, 25)
    with torch.no_grad():
        want = M(z, c).double()                            # the layer-by-layer definition on the CPU (fp32)
    Md = M.to(gpu_device)
    zs, cs = z.to(gpu_device), c.to(gpu_device)
    rows = {}
    with torch.no_grad():
        for n in (1, 4, 5, 8):
            rows[n] = Md(zs[:n], cs[:n])[0].cpu()
    scale = float(want.abs().max())
    for n, r in rows.items():
        assert float((r.double() - want[0]).abs().max()) <= 2e-5 * scale, f'batch {n} vs the CPU definition'
        assert float((r - rows[1]).abs().max()) <= 4e-6 * scale, f'batch {n} vs batch 1: more than rounding'


def test_mapping_per_layer_form_is_bit_equal(gpu_device, tmp_path):
    """`ide3d_mapping` as one launch per layer (what a device takes on which the one-launch kernel's 64 workgroups are not co-resident;
    IDE3D_MAPPING_PER_LAYER=1, read once per process) == the one-launch form, bit for bit: same gemv code, a kernel boundary for each grid barrier."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from training import triplane; from torch_utils import hip_plugin;"
            "torch.manual_seed(0); G = triplane.TriPlaneGenerator().eval().requires_grad_(False).to('cuda:0');"
            "z = torch.from_numpy(np.random.RandomState(1).randn(4, G.z_dim)).float().cuda();"
            "c = triplane.conditioning_label('cuda:0').repeat(4, 1);"
            "torch.set_grad_enabled(False); ws = G.mapping(z, c, truncation_psi=0.7, truncation_cutoff=8); assert hip_plugin.CALLS.get('mapping') == 1;"
            "np.save(sys.argv[1], ws.cpu().numpy())") % (os.path.join(ROOT, 'ide-3d_amd'), ROOT)
    outs = []
    for flag in ('', '1'):
        env = dict(os.environ)
        env.pop('IDE3D_MAPPING_PER_LAYER', None)
        if flag:
            env['IDE3D_MAPPING_PER_LAYER'] = flag
        path = str(tmp_path / f'ws{flag}.npy')
        subprocess.run([sys.executable, '-c', code, path], check=True, env=env, timeout=300)
        outs.append(np.load(path))
    assert np.array_equal(outs[0], outs[1])
