"""Per-kernel roofline table of the hot path at BASELINE config-2 shapes (batch 4, 64x64 rays x 96 samples, 256^2 tri-planes,
512^2 output): one row per hand-written kernel of SURVEY.md §8(a).

    python scripts/kernel_rooflines.py [--iters 20] [--json out.json] [--only substr] [--eager]

For every case: HIP-event time of `iters` launches replayed from one hipGraph (no host launch gaps; `--eager` launches
them one by one instead: that is the mode the rocprofv3 counter passes of scripts/pmc_kernels.sh use), the ALGORITHMIC bytes
or flops of SURVEY.md §8(d) (`(in + out) * sizeof` for streaming ops, `2 * Cin * Cout * k^2 * positions` for convolutions),
and the achieved fraction of the bounding peak (HBM 8.0 TB/s spec, fp32 MFMA 157.3 TFLOP/s; a split-bf16 convolution row is
priced against the dense bf16 MFMA peak 2.5 PFLOP/s divided by its 6 or 3 bf16 products per fp32 product).
`bench.py` embeds the same rows as `roofline_extra`; the committed table lives under profiles/.
"""

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12
HBM_COPY = 6.29e12           # measured copy ceiling (MI355X_MICROARCH.md)
FP32_MFMA_PEAK = 157.3e12
BF16_MFMA_PEAK = 2.5e15      # dense; the split-bf16 convolutions spend 6 (bf16x6) or 3 (bf16x3) bf16 products per fp32 product
ARITH_PRODUCTS = {'bf16x6': 6, 'bf16x3': 3, 'f16x3': 3}
N = 4                        # batch of config 2


def _time(fn, iters, device, eager=False):
    """Average microseconds per call of `fn` (which must only enqueue work on the current stream)."""
    stream = torch.cuda.Stream(device=device)
    stream.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(stream), torch.no_grad():
        for _ in range(2):
            fn()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if eager:
            e0.record(stream)
            for _ in range(iters):
                fn()
            e1.record(stream)
        else:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for _ in range(iters):
                    fn()
            # warm-up to sustained clocks: ~30 ms of continuous load (power management), then one timed replay
            graph.replay()
            stream.synchronize()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.04:
                graph.replay()
                stream.synchronize()
            e0.record(stream)
            graph.replay()
            e1.record(stream)
        stream.synchronize()
    torch.cuda.current_stream(device).wait_stream(stream)
    return e0.elapsed_time(e1) * 1e3 / iters


def cases(device):
    """[(name, kernel-name fragment, bound, algorithmic amount (bytes or flops), callable)]"""
    from dnnlib import util
    from torch_utils import hip_plugin
    from torch_utils.ops import bias_act, filtered_lrelu, upfirdn2d
    from training import distributed_render as dr
    from training import networks, triplane
    from training import volumetric_rendering as vr
    from training import shape_extraction as se

    g = torch.Generator().manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g).to(device)
    out = []

    # ---- a1 bias_act ------------------------------------------------------------------------------------------------
    x = rn(N, 64, 512, 512); b = rn(64)
    out.append(('bias_act lrelu [4,64,512,512] f32', 'bias_act', 'hbm', 2 * x.numel() * 4, lambda: bias_act.bias_act(x, b, act='lrelu')))
    xh = x.half(); bh = b.half()
    out.append(('bias_act lrelu [4,64,512,512] f16', 'bias_act', 'hbm', 2 * xh.numel() * 2, lambda: bias_act.bias_act(xh, bh, act='lrelu')))

    # ---- a2 upfirdn2d -------------------------------------------------------------------------------------------------
    f4 = upfirdn2d.setup_filter([1, 3, 3, 1], device=device)
    for name, shape, kw in (('FIR 4x4 post-tconv 64ch 513->512', (N, 64, 513, 513), dict(padding=[1, 1, 1, 1], gain=4)),
                            ('FIR 4x4 post-tconv 128ch 257->256', (N, 128, 257, 257), dict(padding=[1, 1, 1, 1], gain=4)),
                            ('FIR 4x4 up2 skip 96ch 128->256', (N, 96, 128, 128), dict(up=2, padding=[2, 1, 2, 1], gain=4)),
                            ('FIR 4x4 up2 skip 22ch 256->512', (N, 22, 256, 256), dict(up=2, padding=[2, 1, 2, 1], gain=4))):
        xi = rn(*shape)
        if shape[-1] % 4:       # the (2h + 1)-wide transposed-convolution output arrives with rows padded to a multiple of 4 floats (y_pitch)
            xi = torch.nn.functional.pad(xi, (0, 4 - shape[-1] % 4))[..., :shape[-1]]
        yo = upfirdn2d.upfirdn2d(xi, f4, **kw)
        out.append((name, 'upfirdn2d_tile', 'hbm', (xi.numel() + yo.numel()) * 4, (lambda xi=xi, kw=kw: upfirdn2d.upfirdn2d(xi, f4, **kw))))
    # fused epilogue variant the generator runs after every transposed conv: FIR + noise + bias + lrelu
    xi = torch.nn.functional.pad(rn(N, 64, 513, 513), (0, 3))[..., :513]; nz = rn(512, 512); bb = rn(64)      # padded rows, like the tconv writes them
    upfirdn2d._init()
    out.append(('FIR 4x4 + noise/bias/lrelu 64ch 513->512', 'upfirdn2d_tile', 'hbm', (xi.numel() + N * 64 * 512 * 512) * 4,
                lambda: upfirdn2d._plugin.upfirdn2d_ex(xi, f4, 1, 1, 1, 1, 1, 1, 1, 1, False, 4.0, noise=nz, noise_strength=1.0, bias=bb,
                                                       act=3, alpha=0.2, act_gain=math.sqrt(2), clamp=-1.0)))
    # the last skip accumulation of the backbone, written channels-last (csrc/resample.hip): lo + add in, tri-plane out
    lo_ = rn(N, 96, 128, 128); wide = rn(N, 192, 256, 256)
    out.append(('skip upsample2d + add -> channels-last tri-plane 96ch 128->256', 'skip_upsample_add_cl', 'hbm', (lo_.numel() + 2 * N * 96 * 256 * 256) * 4,
                lambda: hip_plugin.ResamplePlugin.skip_upsample_add_cl(lo_, wide[:, 96:])))
    ft = rn(N, 51, 64, 64)
    out.append(('bilinear 2x split [4,51,64,64] -> 32 + 3 + 19 ch @128', 'bilinear_up2_split', 'hbm', (ft.numel() + N * 54 * 128 * 128) * 4,
                lambda: hip_plugin.ResamplePlugin.bilinear_up2_split(ft, [(0, 32), (0, 3), (32, 19)])))
    # generic kernel: separable 12-tap filter (training/augment.py:295 pattern), up = 2 -> two 1-D passes
    f12 = upfirdn2d.setup_filter([0.0154, 0.0035, -0.1180, -0.0483, 0.4911, 0.7877, 0.4911, -0.0483, -0.1180, 0.0035, 0.0154, 0.0],
                                 device=device, normalize=True)
    xs = rn(N, 64, 128, 128)
    ys = upfirdn2d.upsample2d(xs, f12, up=2)
    mid = N * 64 * 128 * 256        # elements after the first (horizontal) pass
    out.append(('FIR 12-tap separable up2 64ch 128->256 (two 1-D tile passes)', 'upfirdn2d_generic', 'hbm',
                (xs.numel() + 2 * mid + ys.numel()) * 4, lambda: upfirdn2d.upsample2d(xs, f12, up=2)))
    f47 = upfirdn2d.setup_filter(np.outer(np.hanning(47), np.hanning(47)), device=device)
    xv = rn(1, 3, 128, 128)
    yv = upfirdn2d.upsample2d(xv, f47, up=4)
    out.append(('FIR 47x47 up4 3ch 128->512, cell kernel (viewer, viz/renderer.py:360)', 'upfirdn2d_cell', 'hbm',
                (xv.numel() + yv.numel()) * 4, lambda: upfirdn2d.upsample2d(xv, f47, up=4)))

    # ---- a3 filtered_lrelu (StyleGAN3 layer shape, inversion/networks.py:576-597: up 2, down 2, 12-tap separable) -------
    fu = upfirdn2d.setup_filter(np.hanning(14)[1:-1], device=device); fd = upfirdn2d.setup_filter(np.hanning(14)[1:-1], device=device)
    for dt, tag in ((torch.float32, 'f32'), (torch.float16, 'f16')):
        xf = rn(N, 128, 128, 128).to(dt); bf = rn(128).to(dt)
        yf = filtered_lrelu.filtered_lrelu(xf, fu=fu, fd=fd, b=bf, up=2, down=2, padding=[10, 11, 10, 11])
        out.append((f'filtered_lrelu up2 down2 12-tap [4,128,128,128] {tag}', 'filtered_lrelu', 'hbm', (xf.numel() + yf.numel()) * xf.element_size(),
                    (lambda xf=xf, bf=bf: filtered_lrelu.filtered_lrelu(xf, fu=fu, fd=fd, b=bf, up=2, down=2, padding=[10, 11, 10, 11]))))
    xf4 = rn(N, 64, 64, 64); bf4 = rn(64)
    yf4 = filtered_lrelu.filtered_lrelu(xf4, fu=fu, fd=fd, b=bf4, up=4, down=2, padding=[10, 11, 10, 11])
    out.append(('filtered_lrelu up4 down2 12-tap [4,64,64,64] f32', 'filtered_lrelu', 'hbm', (xf4.numel() + yf4.numel()) * 4,
                lambda: filtered_lrelu.filtered_lrelu(xf4, fu=fu, fd=fd, b=bf4, up=4, down=2, padding=[10, 11, 10, 11])))

    # ---- a12 tri-plane gather -----------------------------------------------------------------------------------------
    C, H, M = 32, 256, 64 * 64 * 96
    planes = rn(N, 3 * C, H, H).contiguous(memory_format=torch.channels_last)
    pts, z, d = vr.get_initial_rays_trig(N, 96, device, 18.0, (64, 64), 2.25, 3.3)
    cam = torch.cat([triplane.camera_label(y, device=device) for y in (-0.5, -0.15, 0.2, 0.5)])[:, :16].reshape(-1, 4, 4)
    jit = torch.rand(z.shape, generator=g).to(device)
    wp, zj, dj, *_ = vr.transform_sampled_points(pts, z, d, device, h_stddev=0, v_stddev=0, camera=cam, mode=None, jitter=jit)
    coords = wp.reshape(N, M, 3).contiguous()
    gbytes = (3 * C * H * H + 3 * M + C * M) * 4 * N
    out.append(('tri-plane gather, ray-grid kernel (N=4, 1 tri-plane)', 'triplane_sample_tile', 'hbm', gbytes,
                lambda: util.sample_from_triplane(coords, planes, ray_grid=(64, 64, 96))))
    out.append(('tri-plane gather, flat kernel (N=4, 1 tri-plane)', 'triplane_sample_cl2', 'hbm', gbytes,
                lambda: util.sample_from_triplane(coords, planes, ray_grid=False)))
    # the call exactly as dnnlib/util.py:580 spells it: the ray grid is recognised from the data (first call, cached per M)
    out.append(('tri-plane gather, flat call (no hint: ray grid recognised from the data)', 'triplane_sample_tile', 'hbm', gbytes,
                lambda: util.sample_from_triplane(coords, planes)))

    # ---- a13 / fused renderer -------------------------------------------------------------------------------------------
    torch.manual_seed(0)
    R = triplane.TriplaneRenderer(triplane.GeneratorSpec()).to(device).eval()
    tex = (rn(N, 96, 256, 256) * 0.7).contiguous(memory_format=torch.channels_last)
    geo = (rn(N, 96, 256, 256) * 0.7).contiguous(memory_format=torch.channels_last)
    cam2 = torch.cat([triplane.camera_label(y, device=device) for y in (-0.5, 0.0, 0.5, 0.25)])[:, :16].reshape(-1, 4, 4)
    jit3 = jit.reshape(N, 4096, 96)
    mlp_flops = N * M * 2 * (2 * 32 * 64 + 64 * 20 + 64 * 32)
    fused_bytes = N * (2 * 3 * C * H * H + 4096 * 53 + M) * 4
    # the decoder MLPs run bf16x6 (v_mfma_f32_16x16x32_bf16, 6 products per fp32 product) unless exact fp32 products are selected
    # (v_mfma_f32_16x16x4_f32): the row is priced against the peak of what it runs.  The kernel itself is gather / VALU limited, not
    # MFMA limited (`note`: counters of the last committed rocprofv3 collection).
    mlp_bound = 'mfma' if hip_plugin.conv_arithmetic() == 'fp32' else 'mfma:bf16x6'
    out.append(('render_rays fused (2 gathers + 2 MLPs + compositing), MLP flops', 'render_rays', mlp_bound, mlp_flops,
                lambda: R(tex, geo, cam2, jitter=jit3)))
    out.append(('render_rays fused, compulsory HBM bytes (planes are cache-resident)', 'render_rays', 'hbm', fused_bytes,
                lambda: R(tex, geo, cam2, jitter=jit3)))
    out.append(('sample_voxel [4 x 393216 points] -> [.,52]', 'sample_voxel', 'hbm', N * (2 * 3 * C * H * H + M * 3 + M * 52) * 4,
                lambda: R.sample_voxel(tex, geo, coords)))
    vs = 2.0 / 255
    corner = np.array([-1.0, -1.0, -1.0])
    out.append(('density_lattice 256^3 (1 image, sigma only), geometry-branch MLP flops', 'density_kernel', mlp_bound,
                256 ** 3 * 2 * (32 * 64 + 64 * 1), lambda: R.density_lattice(tex[:1], geo[:1], 256, vs, corner, 0.9, 0, 256 ** 3)))
    out.append(('density_lattice 256^3, HBM bytes (planes + sigma out)', 'density_kernel', 'hbm',
                (3 * C * H * H + 256 ** 3) * 4, lambda: R.density_lattice(tex[:1], geo[:1], 256, vs, corner, 0.9, 0, 256 ** 3)))

    # ---- a14 compositing, a15 sample_pdf ---------------------------------------------------------------------------------
    rs = rn(N, 4096, 96, 52); zz = zj.contiguous(); dd = d.contiguous()
    out.append(('composite [4,4096,96,52]', 'composite', 'hbm', N * 4096 * (96 * 53 + 51 + 1 + 96) * 4,
                lambda: vr.fancy_integration(rs, dd, zz, device, noise_std=0, clamp_mode='softplus')))
    wts = torch.rand(N * 4096, 94, generator=g).to(device) + 1e-5
    bins = torch.sort(torch.rand(N * 4096, 95, generator=g), dim=1).values.to(device)
    u = torch.rand(N * 4096, 96, generator=g).to(device)
    out.append(('sample_pdf 16384 rays x 94 bins x 96 draws', 'sample_pdf', 'hbm', N * 4096 * (95 + 94 + 96 + 96) * 4,
                lambda: vr.sample_pdf(bins, wts, 96, det=False, u=u)))

    # ---- f1 frame conversion ---------------------------------------------------------------------------------------------
    img = rn(N, 3, 512, 512); seg = rn(N, 19, 512, 512); pal = dr.palette_tensor(19, device)
    out.append(('frame_u8 [4,3+19,512,512] -> uint8 [4,512,1024,3]', 'frame_u8', 'hbm', N * 512 * 512 * (22 * 4 + 6), lambda: dr.frames_u8(img, seg, pal)))

    # ---- a7 modulated convolutions (every mode) ----------------------------------------------------------------------------
    mc = hip_plugin.ModconvPlugin.modconv2d

    ARITH_CODE = {'fp32': 1, 'bf16x3': 3, 'bf16x6': 6, 'f16x3': 16}

    def conv_case(tag, cin, cout, res, k=3, mode=0, ariths=('f16x3', 'bf16x6', 'fp32')):
        """One row per arithmetic for the layers that have a choice (3x3 stride 1 and transposed); the bound of a split-bf16 row is
        the bf16 MFMA peak / its products per fp32 product ('mfma:bf16x6' = 416.7 TFLOP/s of fp32-equivalent work)."""
        xx = rn(N, cin, res, res); ww = rn(cout, cin, k, k); ss = rn(N, cin) + 1; dc = torch.rand(N, cout, generator=g).to(device)
        xam = xx.abs().amax(dim=(1, 2, 3))[:, None].repeat(1, 32 * 64).contiguous()          # what the producing kernel records on the render path (f16x3)
        if mode == 1:
            bz = rn(cout)
            fn = lambda: mc(xx, ww, None, None, None, 0.0, bz, 3, 0.2, math.sqrt(2), -1.0, mode=1)
            pos = ((res - 3) // 2 + 1) ** 2
            out.append((tag, 'modconv_kernel', 'mfma', 2 * cin * cout * k * k * pos * N, fn))
            return
        for ar in ariths:
            code = ARITH_CODE[ar]
            if mode == 0:
                nzz = rn(res, res); bz = rn(cout)
                fn = lambda code=code, nzz=nzz, bz=bz: mc(xx, ww, ss, dc, nzz, 1.0, bz, 3, 0.2, math.sqrt(2), -1.0, arith=code, x_amax=xam)
            else:
                fn = lambda code=code: mc(xx, ww, ss, dc, None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=2, arith=code, x_amax=xam)
            kern = 'modconv_kernel' if ar == 'fp32' else 'modconv_split_kernel'
            out.append((f'{tag} [{ar}]', kern, 'mfma' if ar == 'fp32' else f'mfma:{ar}', 2 * cin * cout * k * k * res * res * N, fn))

    conv_case('modconv 3x3 128->128 @256', 128, 128, 256)
    conv_case('modconv 3x3 256->256 @128', 256, 256, 128)
    conv_case('modconv 3x3 512->512 @64', 512, 512, 64)
    conv_case('modconv 3x3 64->64 @512', 64, 64, 512)
    conv_case('modconv transposed 3x3 512->256 in@64', 512, 256, 64, mode=2)
    conv_case('modconv transposed 3x3 256->128 in@128', 256, 128, 128, mode=2)
    conv_case('modconv transposed 3x3 128->64 in@256', 128, 64, 256, mode=2)
    conv_case('modconv transposed 3x3 32->128 in@128 (b256.conv0)', 32, 128, 128, mode=2)
    # the low-resolution half of the backbone (4^2 .. 32^2 at batch 4: a few dozen workgroups + split-K each)
    for res in (32, 16, 8, 4):
        conv_case(f'modconv 3x3 512->512 @{res} (low-res backbone)', 512, 512, res)
    for res in (32, 16, 8, 4):
        conv_case(f'modconv transposed 3x3 512->512 in@{res} (low-res backbone)', 512, 512, res, mode=2)
    # the low-resolution block group (csrc/lowres.hip, round 6): the 4^2 .. 16^2 (batch 4) / .. 32^2 (batch 1) layers of the backbone as one
    # launch per phase, against the per-layer launches it replaces.  flops = the covered 3x3 / transposed layers + heads
    if hip_plugin.conv_arithmetic() in ('bf16x6', 'bf16x3'):
        from training import networks as _nw
        torch.manual_seed(0)
        Gs = triplane.TriPlaneGenerator().eval().requires_grad_(False).to(device).synthesis
        blocks = [getattr(Gs, f'vb{r}') for r in Gs.voxel_block_resolutions]
        for nb in (N, 1):
            wsb = Gs.split_ws(rn(nb, Gs.num_ws, 512))[0]
            with torch.no_grad():
                got = _nw.lowres_group_forward(blocks, wsb, noise_mode='const')
            if got is None:
                continue
            nblk, resume = got[3], got[4]
            fl = 0
            for bi in range(nblk + (1 if resume else 0)):
                r = 4 << bi
                if bi > 0:
                    fl += 2 * 512 * 512 * 9 * (r // 2) ** 2 * nb                      # transposed 3x3: 9 taps per INPUT position
                if bi < nblk:
                    fl += 2 * 512 * 512 * 9 * r * r * nb + 2 * 512 * 192 * r * r * nb      # conv1 + both heads
            prod = ARITH_PRODUCTS.get(hip_plugin.conv_arithmetic(), 6)
            for tag, env in (('group', None), ('per-layer launches', '1')):
                def fn(wsb=wsb, env=env, nblk=nblk, resume=resume):
                    if env: os.environ['IDE3D_NO_LOWRES_GROUP'] = env
                    try:
                        x = img = seg = None
                        grp = _nw.lowres_group_forward(blocks, wsb, noise_mode='const')
                        start, res_ = (grp[3], grp[4]) if grp else (0, False)
                        if grp: x, img, seg = grp[:3]
                        for i in range(start, nblk + (1 if resume else 0)):
                            if i < nblk:
                                x, img, seg = blocks[i](x, img, wsb[i], condition_img=seg, noise_mode='const', **(dict(_resume_after_conv0=True) if (res_ and i == start) else {}))
                            elif not grp:
                                x = blocks[i].conv0(x, wsb[i][:, 0], noise_mode='const')
                    finally:
                        os.environ.pop('IDE3D_NO_LOWRES_GROUP', None)
                out.append((f'low-res block group 4^2..{4 << (nblk - 1)}^2{"+up" if resume else ""} batch {nb}: {tag} [{hip_plugin.conv_arithmetic()}]',
                            'lr::phase_kernel' if env is None else 'modconv_split', f'mfma:{hip_plugin.conv_arithmetic()}', fl, fn))
    conv_case('conv 3x3 stride 2 64->128 in@257 (encoder)', 64, 128, 257, mode=1)
    # the face parser of the editing loop (training/face_parsing.py): 32 folded conv+BN launches on the fp32 MFMA kernel + ATen glue, 512 x 512
    from training import face_parsing as _fp
    torch.manual_seed(1)
    Pn = _fp.BiSeNet(20).eval().requires_grad_(False).to(device)
    x_face = rn(1, 3, 512, 512)          # (a NEW name: the lambdas above close over `xi`, `nz`, `bb` by name)
    out.append(('face parser BiSeNet 512x512 batch 1 (ResNet-18 context path: ~19 GFLOP)', 'modconv_kernel', 'mfma', 19.1e9, lambda: Pn(x_face)))
    for cin, cout, res in ((128, 192, 256), (256, 192, 128), (64, 22, 512), (128, 22, 256)):
        xx = rn(N, cin, res, res); ww = rn(N, cout, cin, 1, 1); bz = rn(cout)
        out.append((f'dual head 1x1 (per-image weights) {cin}->{cout} @{res}', 'modconv_kernel', 'hbm', (cin + cout) * res * res * N * 4,
                    (lambda xx=xx, ww=ww, bz=bz: mc(xx, ww, None, None, None, 0.0, bz, 1, 0.0, 1.0, 256.0, arith=16))))      # heads under f16x3 = the bf16x6 head kernel

    # ---- a8 style / demodulation / head folding / mapping -------------------------------------------------------------------
    networks._style_init()
    sp = networks._style_plugin
    w = rn(N, 512); aw = rn(512, 512); ab = rn(512); wsq = torch.rand(512, 512, generator=g).to(device)
    out.append(('style affine + demodulation 512->512 (GEMV pair)', 'style_', 'hbm', (aw.numel() + wsq.numel()) * 4,
                lambda: sp.style_demod(w, aw, ab, 1 / math.sqrt(512), 1.0, wsq)))
    torch.manual_seed(0)
    Mnet = networks.MappingNetwork(512, 25, 512, 18).to(device).eval()
    zz_ = rn(N, 512); cc_ = triplane.conditioning_label(device).repeat(N, 1)
    map_bytes = sum(p.numel() for p in Mnet.parameters()) * 4
    out.append(('mapping network z,c -> ws [4,18,512] (weight bytes)', 'mapping', 'hbm', map_bytes, lambda: Mnet(zz_, cc_)))
    return out


def _pmc_notes():
    """{kernel-name fragment: 'mfma_busy x, VALU active y, ...'} from the newest committed profiles/round*/render_pmc.json + modconv_pmc.json
    (counters are NOT measured in this run: rocprofv3 --pmc passes are separate, scripts/pmc_kernels.sh)."""
    import glob
    notes = {}
    for fname in ('render_pmc.json', 'modconv_pmc.json'):
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*', fname)))
        if not files:
            continue
        try:
            kernels = json.load(open(files[-1])).get('kernels', {})
        except (OSError, ValueError):
            continue
        src = os.path.relpath(files[-1], ROOT)
        for kname, c in kernels.items():
            if 'mfma_busy_frac' in c:
                notes[kname] = dict(mfma_busy_frac=round(c['mfma_busy_frac'], 3),
                                    valu_active_frac_of_wave_cycles=round(c.get('sq_active_inst_valu_frac_of_wave_cycles', float('nan')), 3),
                                    sq_wait_any_frac_of_wave_cycles=round(c.get('sq_wait_any_frac_of_wave_cycles', float('nan')), 3), source=src)
    return notes


def measure_all(device, iters=20, only=None, eager=False):
    rows = []
    notes = _pmc_notes()
    for name, kernel, bound, amount, fn in cases(device):
        if only and only not in name and only not in kernel:
            continue
        us = _time(fn, iters, device, eager=eager)
        rate = amount / (us * 1e-6)
        if bound == 'hbm':
            rows.append(dict(name=name, kernel=kernel, bound='hbm', us=us, algorithmic_bytes=amount, achieved=rate / 1e9, peak=HBM_PEAK / 1e9,
                             unit='GB/s', frac=rate / HBM_PEAK, frac_of_copy_ceiling=rate / HBM_COPY))
        elif bound == 'mfma':
            rows.append(dict(name=name, kernel=kernel, bound='mfma', us=us, algorithmic_flops=amount, achieved=rate / 1e12, peak=FP32_MFMA_PEAK / 1e12,
                             unit='TFLOP/s', frac=rate / FP32_MFMA_PEAK))
        else:
            # split-bf16 convolution: fp32-equivalent flops against the bf16 MFMA peak / products per fp32 product
            peak = BF16_MFMA_PEAK / ARITH_PRODUCTS[bound.split(':')[1]]
            rows.append(dict(name=name, kernel=kernel, bound=bound, us=us, algorithmic_flops=amount, achieved=rate / 1e12, peak=peak / 1e12,
                             unit='TFLOP/s', frac=rate / peak, frac_of_fp32_mfma_peak=rate / FP32_MFMA_PEAK))
        if kernel in ('render_rays', 'density_kernel'):
            hit = [v for k, v in notes.items() if k.startswith(kernel)]
            if hit:
                rows[-1]['counters_not_measured_in_this_run'] = hit[0]
                rows[-1]['limiter'] = 'tri-plane gathers + vector instructions (matrix pipe mostly idle): the MLP-flops fraction is an upper bound on nothing'
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--json', default=None)
    ap.add_argument('--only', default=None)
    ap.add_argument('--eager', action='store_true')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    from torch_utils import hip_plugin
    hip_plugin.load()
    rows = measure_all(dev, args.iters, args.only, args.eager)
    print(f'{"kernel case":78s} {"us":>9s} {"achieved":>12s} {"% of peak":>10s}')
    for r in rows:
        print(f'{r["name"]:78s} {r["us"]:9.1f} {r["achieved"]:9.1f} {r["unit"]:>8s} {100 * r["frac"]:9.1f}%')
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        json.dump(rows, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
