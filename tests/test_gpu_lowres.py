"""The low-resolution block group (csrc/lowres.hip, `ide3d_lowres_group`) against the per-layer HIP path it replaces and against a float64
definition of the same blocks (reference semantics: inversion/networks.py:966-1139 `SegSynthesisBlock`, :330-514 `SynthesisLayer`,
:670-713 `ToRGBLayer`; conv2d_resample.py:112-129 for the up-sampling layers)."""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _blocks(C, nblocks, w_dim, device, img_ch=12, seg_ch=8, conv_clamp=None, seed=0):
    from training import triplane
    torch.manual_seed(seed)
    blocks = []
    for i in range(nblocks):
        res = 4 << i
        b = triplane.VoxelBlock(0 if i == 0 else C, C, w_dim=w_dim, resolution=res, img_channels=img_ch, seg_channels=seg_ch, is_last=False,
                                architecture='skip', conv_clamp=conv_clamp, layer_name='training.networks.SynthesisLayer')
        for lay in ([b.conv1] if i == 0 else [b.conv0, b.conv1]):
            lay.noise_strength.data.fill_(0.37)
            lay.bias.data.normal_(0, 0.3)
        b.torgb.bias.data.normal_(0, 0.2)
        b.toseg.bias.data.normal_(0, 0.2)
        blocks.append(b.eval().requires_grad_(False).to(device))
    return blocks


def _split(blocks, ws):
    out, idx = [], 0
    for b in blocks:
        out.append(ws.narrow(1, idx, b.num_conv + b.num_torgb))
        idx += b.num_conv
    return out


def _run(blocks, ws_list, group, persistent=None, start_state=None):
    """-> (x, img, seg) after all blocks, with / without the group launch; also how many blocks the group covered"""
    from training import networks
    old = {k: os.environ.get(k) for k in ('IDE3D_NO_LOWRES_GROUP', 'IDE3D_LOWRES_PERSISTENT')}
    try:
        os.environ.pop('IDE3D_NO_LOWRES_GROUP', None)
        if not group:
            os.environ['IDE3D_NO_LOWRES_GROUP'] = '1'
        if persistent is not None:
            os.environ['IDE3D_LOWRES_PERSISTENT'] = '1' if persistent else '0'
        x = img = seg = None
        start, resume, info = 0, False, None
        with torch.no_grad():
            grp = networks.lowres_group_forward(blocks, ws_list, noise_mode='const')
            if grp is not None:
                x, img, seg, start, resume = grp
                info = (start, resume, x.clone(), img.clone(), seg.clone())
            for i, (b, w) in enumerate(zip(blocks, ws_list)):
                if i < start:
                    continue
                extra = dict(_resume_after_conv0=True) if (resume and i == start) else {}
                x, img, seg = b(x, img, w, condition_img=seg, noise_mode='const', **extra)
        return x, img, seg, info
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('C,nblocks,n', [(64, 3, 1), (64, 4, 3), (512, 4, 1), (512, 3, 4), (512, 3, 2), (128, 2, 8)])
@pytest.mark.parametrize('persistent', [False, True])
def test_group_equals_per_layer_path(gpu_device, C, nblocks, n, persistent):
    from torch_utils import hip_plugin
    assert hip_plugin.conv_arithmetic() == 'bf16x6'
    w_dim = 64
    blocks = _blocks(C, nblocks, w_dim, gpu_device)
    ws = torch.randn([n, sum(b.num_conv for b in blocks) + 1, w_dim], device=gpu_device)
    ws_list = _split(blocks, ws)
    before = hip_plugin.CALLS.get('lowres_group', 0)
    xg, ig, sg, info = _run(blocks, ws_list, True, persistent)
    assert info is not None, 'the group launch did not apply'
    assert hip_plugin.CALLS.get('lowres_group', 0) == before + 1
    xr, ir, sr, none = _run(blocks, ws_list, False)
    assert none is None
    # same arithmetic (bf16x6 products, fp32 accumulation), different summation order: a few ulp of the tensor's scale
    assert _rel(xg, xr) < 1e-5, _rel(xg, xr)
    assert _rel(ig, ir) < 1e-5 and _rel(sg, sr) < 1e-5, (_rel(ig, ir), _rel(sg, sr))
    assert hip_plugin.exclusive_violations() == (0, '')


def test_group_with_clamp_and_in_bf16x3(gpu_device):
    """conv_clamp active in every layer and head (the fp16-block setting of a released pickle, inversion/networks.py:1058-1060), and the two-piece
    arithmetic (bf16x3: PARTS = 2 instantiation of every phase) — each against the per-layer path in the same setting."""
    from torch_utils import hip_plugin
    for arith, tol in (('bf16x6', 1e-5), ('bf16x3', 2e-4)):
        hip_plugin.conv_arithmetic(arith)
        try:
            blocks = _blocks(512, 3, 64, gpu_device, conv_clamp=0.6, seed=3)
            ws_list = _split(blocks, torch.randn([2, sum(b.num_conv for b in blocks) + 1, 64], device=gpu_device))
            xg, ig, sg, info = _run(blocks, ws_list, True)
            assert info is not None
            assert float(xg.abs().max()) <= 0.6 * 1.0000001 and float(info[3].abs().max()) > 0.3          # the clamp bites (act gain sqrt 2 on unit-variance data)
            xr, ir, sr, _ = _run(blocks, ws_list, False)
            assert _rel(xg, xr) < tol and _rel(ig, ir) < tol and _rel(sg, sr) < tol, (arith, _rel(xg, xr), _rel(ig, ir), _rel(sg, sr))
        finally:
            hip_plugin.conv_arithmetic('default')


def test_group_outputs_at_its_own_boundary(gpu_device):
    """What leaves the group (x in front of the next block / conv0's output inside it, the skip images) against the per-layer path cut at the
    same place — a whole-backbone tolerance would hide an O(1) error of a sub-stage behind later layers."""
    from training import networks
    C, w_dim = 512, 64
    for n, nblocks in ((1, 5), (4, 4)):
        blocks = _blocks(C, nblocks, w_dim, gpu_device, seed=1)
        ws = torch.randn([n, sum(b.num_conv for b in blocks) + 1, w_dim], device=gpu_device)
        ws_list = _split(blocks, ws)
        _, _, _, info = _run(blocks, ws_list, True)
        start, resume, xg, ig, sg = info
        assert start >= 2
        os.environ['IDE3D_NO_LOWRES_GROUP'] = '1'
        try:
            with torch.no_grad():
                x = img = seg = None
                for i in range(start):
                    x, img, seg = blocks[i](x, img, ws_list[i], condition_img=seg, noise_mode='const')
                if resume:
                    x = blocks[start].conv0(x, ws_list[start][:, 0], noise_mode='const')
        finally:
            os.environ.pop('IDE3D_NO_LOWRES_GROUP')
        assert xg.shape == x.shape and ig.shape == img.shape and sg.shape == seg.shape
        assert _rel(xg, x) < 5e-6 and _rel(ig, img) < 5e-6 and _rel(sg, seg) < 5e-6, (n, _rel(xg, x), _rel(ig, img), _rel(sg, seg))


def test_group_against_float64_definition(gpu_device):
    """The blocks' mathematics in float64 (modulate, 3x3 conv / transposed conv + FIR, demodulate, noise, bias, lrelu, heads, skip up-sampling)."""
    import torch.nn.functional as F
    C, w_dim, n = 64, 32, 1          # (batch 1: all three blocks fit the group)
    blocks = _blocks(C, 3, w_dim, gpu_device, seed=2)
    ws = torch.randn([n, sum(b.num_conv for b in blocks) + 1, w_dim], device=gpu_device)
    ws_list = _split(blocks, ws)
    xg, ig, sg, info = _run(blocks, ws_list, True)
    assert info is not None and info[0] == 3

    def layer(lay, x, w, up):
        wt = lay.weight.double()
        s = (w.double() @ lay.affine.weight.double().t()) * lay.affine.weight_gain + lay.affine.bias.double() * lay.affine.bias_gain
        wm = wt[None] * s[:, None, :, None, None]
        d = (wm.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
        wm = wm * d[:, :, None, None, None]
        ys = []
        for i in range(x.shape[0]):
            if up == 2:
                y = F.conv_transpose2d(x[i:i + 1], wm[i].transpose(0, 1), stride=2)
                f = lay.resample_filter.double()
                y = F.pad(y, [1, 1, 1, 1])
                y = F.conv2d(y, (f.flip([0, 1]) * 4)[None, None].repeat(y.shape[1], 1, 1, 1), groups=y.shape[1])
            else:
                y = F.conv2d(x[i:i + 1], wm[i], padding=1)
            ys.append(y)
        y = torch.cat(ys) + (lay.noise_const * lay.noise_strength).double() + lay.bias.double()[None, :, None, None]
        return F.leaky_relu(y, 0.2) * lay.act_gain

    def head(t, x, w):
        s = ((w.double() @ t.affine.weight.double().t()) * t.affine.weight_gain + t.affine.bias.double() * t.affine.bias_gain) * t.weight_gain
        y = torch.einsum('oc,nc,nchw->nohw', t.weight.double()[:, :, 0, 0], s, x)
        return y + t.bias.double()[None, :, None, None]

    x = img = seg = None
    for bi, (b, w) in enumerate(zip(blocks, ws_list)):
        if bi == 0:
            x = b.const.double()[None].expand(n, -1, -1, -1)
            x = layer(b.conv1, x, w[:, 0], 1)
        else:
            x = layer(b.conv0, x, w[:, 0], 2)
            x = layer(b.conv1, x, w[:, 1], 1)
        wh = w[:, b.num_conv]
        yi, ys = head(b.torgb, x, wh), head(b.toseg, x, wh)
        if img is not None:
            img = _up2_f64(img, b.resample_filter.double())
            seg = _up2_f64(seg, b.resample_filter.double())
            img, seg = img + yi, seg + ys
        else:
            img, seg = yi, ys
    assert _rel(xg, x) < 4e-6 and _rel(ig, img) < 4e-6 and _rel(sg, seg) < 4e-6, (_rel(xg, x), _rel(ig, img), _rel(sg, seg))


def _up2_f64(x, f):
    """upsample2d(x, f) (upfirdn2d.py:313-349) in float64: zero insertion x 2, pad (2, 1), true convolution with f, gain 4"""
    import torch.nn.functional as F
    n, c, h, w = x.shape
    xu = torch.zeros([n, c, 2 * h, 2 * w], dtype=x.dtype, device=x.device)
    xu[:, :, ::2, ::2] = x
    xu = F.pad(xu, [2, 1, 2, 1])
    return F.conv2d(xu, (f.flip([0, 1]) * 4)[None, None].repeat(c, 1, 1, 1), groups=c)


def test_backbone_uses_the_group_and_replays_bit_equal(gpu_device):
    """Full-size backbone: the group launch is taken (batch 1 and 4), agrees with the per-layer path, and a hipGraph replay of it is bit-equal
    to the eager launches (persistent and per-phase forms)."""
    from torch_utils import hip_plugin
    from training import triplane, graph_cache
    torch.manual_seed(0)
    G = triplane.TriPlaneGenerator().eval().requires_grad_(False).to(gpu_device)
    syn = G.synthesis
    for n in (1, 4):
        ws = torch.randn([n, G.num_ws, G.w_dim], device=gpu_device)
        for persistent in ('1', '0'):
            os.environ['IDE3D_LOWRES_PERSISTENT'] = persistent
            try:
                graph_cache.reset(syn)
                graph_cache.STATS.clear()
                before = hip_plugin.CALLS.get('lowres_group', 0)
                with graph_cache.disabled():
                    a = syn.planes(ws)
                assert hip_plugin.CALLS.get('lowres_group', 0) == before + 1
                os.environ['IDE3D_NO_LOWRES_GROUP'] = '1'
                with graph_cache.disabled():
                    b = syn.planes(ws)
                os.environ.pop('IDE3D_NO_LOWRES_GROUP')
                for u, v in zip(a, b):
                    assert _rel(u, v) < 1e-5
                outs = [syn.planes(ws) for _ in range(4)]          # eager, capture, replay, replay
                assert graph_cache.STATS['replay'] >= 2
                for o in outs:
                    for u, v in zip(o, a):
                        assert torch.equal(u, v)
            finally:
                os.environ.pop('IDE3D_NO_LOWRES_GROUP', None)
                os.environ.pop('IDE3D_LOWRES_PERSISTENT', None)


def test_hooks_and_other_arithmetics_keep_the_per_layer_path(gpu_device):
    from torch_utils import hip_plugin
    from training import networks
    blocks = _blocks(64, 2, 32, gpu_device)
    ws_list = _split(blocks, torch.randn([1, 4, 32], device=gpu_device))
    with torch.no_grad():
        assert networks.lowres_group_forward(blocks, ws_list, noise_mode='const') is not None
        assert networks.lowres_group_forward(blocks, ws_list, noise_mode='random') is None
        h = blocks[1].conv0.register_forward_hook(lambda m, i, o: None)
        got = networks.lowres_group_forward(blocks, ws_list, noise_mode='const')
        h.remove()
        assert got is None or got[3] == 1          # the hooked block stays out of the group
        try:
            hip_plugin.conv_arithmetic('fp32')
            assert networks.lowres_group_forward(blocks, ws_list, noise_mode='const') is None
        finally:
            hip_plugin.conv_arithmetic('default')
