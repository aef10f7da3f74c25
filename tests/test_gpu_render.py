"""Parity of the fused ray-marcher and of the whole generator on the GPU against the reference-generated golden
vectors (tiny configuration) and the CPU oracle (full `ide3d-ffhq-64-512` configuration).  `pytest -m gpu`.

Tolerances: renderer outputs 2e-4 relative to the feature scale (fp32 MFMA MLP + fast exp/log in softplus);
full generator images 2e-3 relative to the image scale (hundreds of fp32 convolutions with different summation order).
"""

import math

import numpy as np
import pytest
import torch

from oracle import fast_ops
from oracle import generator as ogen
from oracle import ops as oracle_ops
from oracle import spec as ospec
from util import assert_close, t

pytestmark = pytest.mark.gpu


def _calls(name):
    from torch_utils import hip_plugin
    return hip_plugin.CALLS.get(name, 0)


def _rel(actual, expected, tol, what):
    a = actual.detach().cpu().double(); e = torch.as_tensor(expected).detach().cpu().double()
    scale = float(e.abs().max()) + 1e-12
    err = float((a - e).abs().max())
    assert err <= tol * scale, f'{what}: max abs err {err:.3e} > {tol} * scale {scale:.3e}'


def _load(golden, device):
    from training import triplane
    (cfg, a), = golden('generator_tiny').cases
    G = triplane.TriPlaneGenerator(triplane.tiny_spec()).eval()
    G.load_state_dict({k[len('sd_'):]: t(v) for k, v in a.items() if k.startswith('sd_')})
    return G.to(device), cfg, a


def test_fused_renderer_tiny_vs_reference(golden, gpu_device):
    from torch_utils import hip_plugin
    hip_plugin.CALLS.clear()
    G, cfg, a = _load(golden, gpu_device)
    cam = t(a['in_c'], gpu_device)[:, :16].reshape(-1, 4, 4)
    with torch.no_grad():
        feat, depth, wsum = G.synthesis.renderer(t(a['out_img_v'], gpu_device), t(a['out_seg_v'], gpu_device), cam,
                                                 jitter=t(a['in_jitter'], gpu_device))
    assert _calls('render_rays') == 1, 'the fused HIP kernel must have run'
    _rel(feat, a['out_feat'], 2e-4, 'composited features')
    _rel(depth, a['out_depth'], 1e-4, 'depth')
    with torch.no_grad():
        sv = G.synthesis.renderer.sample_voxel(t(a['out_img_v'], gpu_device), t(a['out_seg_v'], gpu_device), t(a['in_sample_pts'], gpu_device))
        sig = G.synthesis.renderer.sample_voxel(t(a['out_img_v'], gpu_device), t(a['out_seg_v'], gpu_device), t(a['in_sample_pts'], gpu_device),
                                                sigma_only=True)
    assert _calls('sample_voxel') == 2
    _rel(sv, a['out_sample_out'], 2e-4, 'sample_voxel')
    _rel(sig, a['out_sample_out'][:, -1], 2e-4, 'sample_voxel sigma_only')


def test_generator_tiny_vs_reference(golden, gpu_device):
    G, cfg, a = _load(golden, gpu_device)
    with torch.no_grad():
        ws = G.mapping(t(a['in_z'], gpu_device), t(a['in_c_cond'], gpu_device), truncation_psi=cfg['truncation_psi'])
        assert_close(ws, a['out_ws'], rtol=1e-4, atol=1e-5, what='ws')
        out = G.synthesis(t(a['out_ws'], gpu_device), c=t(a['in_c'], gpu_device), noise_mode='const',
                          ray_jitter=t(a['in_jitter'], gpu_device), return_dict=True)
    _rel(out['planes'][0], a['out_img_v'], 2e-4, 'texture tri-plane')
    _rel(out['planes'][1], a['out_seg_v'], 2e-4, 'semantic tri-plane')
    _rel(out['image_depth'], a['out_depth'], 2e-4, 'depth')
    _rel(out['image'], a['out_img'], 1e-3, 'image')
    _rel(out['image_seg'], a['out_seg'], 1e-3, 'seg')
    assert _calls('modconv2d') > 10 and _calls('upfirdn2d') > 5 and _calls('render_rays') >= 1


def test_fused_renderer_full_size_vs_stepwise_and_oracle(gpu_device):
    """C=32, hidden 64, 64x64 rays, 96 steps: fused kernel vs the step-wise HIP ops and (a ray subset) the CPU oracle."""
    from training import triplane
    from training import volumetric_rendering as vr
    from dnnlib import util
    torch.manual_seed(0)
    sp = triplane.GeneratorSpec()
    R = triplane.TriplaneRenderer(sp).to(gpu_device).eval()
    with torch.no_grad():
        for p in R.parameters():
            if p.ndim == 1:
                p.copy_(torch.randn_like(p) * 0.2)
    g = torch.Generator().manual_seed(1)
    n = 2
    tex = (torch.randn(n, 96, 256, 256, generator=g) * 0.7).to(gpu_device)
    geo = (torch.randn(n, 96, 256, 256, generator=g) * 0.7).to(gpu_device)
    cam = torch.cat([triplane.camera_label(0.4), triplane.camera_label(-0.3, pitch=1.4)])[:, :16].reshape(-1, 4, 4).to(gpu_device)
    jit = torch.rand(n, 4096, 96, generator=g).to(gpu_device)
    with torch.no_grad():
        feat, depth, wsum = R(tex, geo, cam, jitter=jit)
        # step-wise on the GPU: reference-shaped pipeline on the individual HIP ops
        pts, z, d = vr.get_initial_rays_trig(n, 96, gpu_device, sp.fov, (64, 64), sp.ray_start, sp.ray_end)
        wp, z, _, _, _, _ = vr.transform_sampled_points(pts, z, d, gpu_device, h_stddev=0, v_stddev=0, camera=cam, mode=None, jitter=jit.unsqueeze(-1))
        flat = wp.reshape(n, -1, 3)
        out = R.decoder(util.sample_from_triplane(flat, tex), util.sample_from_triplane(flat, geo)).reshape(n, 4096, 96, -1)
        f2, d2, w2 = vr.fancy_integration(out, d, z, gpu_device, noise_std=0, clamp_mode='softplus')
    f2 = f2.permute(0, 2, 1).reshape(n, -1, 64, 64)
    _rel(feat, f2, 3e-4, 'fused vs step-wise features')
    _rel(depth, d2.permute(0, 2, 1).reshape(n, 1, 64, 64), 1e-4, 'fused vs step-wise depth')
    _rel(wsum, w2.sum(2).permute(0, 2, 1).reshape(n, 1, 64, 64), 1e-4, 'weight sum')
    # oracle on 64 rays of image 0
    sd = {'synthesis.renderer.' + k: v.detach().cpu() for k, v in R.state_dict().items()}
    osp = ospec.Spec()
    rays = torch.arange(0, 4096, 64)
    p0, z0, d0 = fast_ops.initial_rays(1, 96, osp.fov, (64, 64), osp.ray_start, osp.ray_end)
    p0, z0, d0 = p0[:, rays], z0[:, rays], d0[:, rays]
    p0, z0 = fast_ops.perturb(p0, z0, d0, jit[:1, rays].cpu().unsqueeze(-1))
    w0 = fast_ops.to_world(p0, cam[:1].cpu())
    o = ogen.sample_voxel(sd, osp, tex[:1].cpu(), geo[:1].cpu(), w0.reshape(1, -1, 3), fast_ops).reshape(1, len(rays), 96, -1)
    fo, do, _ = fast_ops.composite(o, d0, z0)
    got = feat[0].reshape(51, 4096)[:, rays].t()
    _rel(got, fo[0], 3e-4, 'fused vs oracle features')
    _rel(depth[0].reshape(4096)[rays], do[0, :, 0], 1e-4, 'fused vs oracle depth')


def test_hierarchical_pass_gpu(golden, gpu_device):
    """Importance pass on the HIP ops (sample_voxel, composite, sample_pdf kernels + torch sort) vs the CPU oracle: tiny
    configuration from the reference-built fixture, and one full-size image (64 x 64 rays, 96 + 96 samples)."""
    from torch_utils import hip_plugin
    from training import triplane
    G, cfg, a = _load(golden, gpu_device)
    sp = G.synthesis.renderer.spec
    rays, steps = sp.render_size ** 2, sp.num_steps
    u = torch.rand(2 * rays, steps, generator=torch.Generator().manual_seed(5))
    cam = t(a['in_c'])[:, :16].reshape(-1, 4, 4)
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    want = ogen.render(sd, ospec.tiny(), t(a['out_img_v']), t(a['out_seg_v']), cam, jitter=t(a['in_jitter']), hierarchical=True, importance_u=u)
    hip_plugin.CALLS.clear()
    with torch.no_grad():
        got = G.synthesis.renderer(t(a['out_img_v'], gpu_device), t(a['out_seg_v'], gpu_device), cam.to(gpu_device),
                                   jitter=t(a['in_jitter'], gpu_device), hierarchical=True, importance_u=u.to(gpu_device))
    assert _calls('sample_pdf') == 1 and _calls('sample_voxel') == 2 and _calls('composite') == 2 and _calls('render_rays') == 0
    _rel(got[0], want[0], 3e-4, 'hierarchical features (tiny)')
    _rel(got[1], want[1], 1e-4, 'hierarchical depth (tiny)')
    # full size, one image
    torch.manual_seed(0)
    fsp = triplane.GeneratorSpec()
    R = triplane.TriplaneRenderer(fsp).eval()
    with torch.no_grad():
        for p in R.parameters():
            if p.ndim == 1:
                p.copy_(torch.randn_like(p) * 0.2)
    g = torch.Generator().manual_seed(3)
    tex, geo = torch.randn(1, 96, 256, 256, generator=g) * 0.7, torch.randn(1, 96, 256, 256, generator=g) * 0.7
    cam = triplane.camera_label(0.3)[:, :16].reshape(-1, 4, 4)
    jit, u = torch.rand(1, 4096, 96, generator=g), torch.rand(4096, 96, generator=g)
    sd = {'synthesis.renderer.' + k: v.detach() for k, v in R.state_dict().items()}
    want = ogen.render(sd, ospec.Spec(), tex, geo, cam, jitter=jit, ops=fast_ops, hierarchical=True, importance_u=u)
    R = R.to(gpu_device)
    with torch.no_grad():
        got = R(tex.to(gpu_device), geo.to(gpu_device), cam.to(gpu_device), jitter=jit.to(gpu_device), hierarchical=True,
                importance_u=u.to(gpu_device))
    _rel(got[0], want[0], 5e-4, 'hierarchical features (full size)')
    _rel(got[1], want[1], 2e-4, 'hierarchical depth (full size)')
    _rel(got[2], want[2], 2e-4, 'hierarchical weight sum (full size)')


def test_generator_full_size_vs_oracle(gpu_device):
    """Random-init ide3d-ffhq-64-512 generator, one image: HIP path vs the CPU oracle (fp32 torch formulation)."""
    from training import triplane
    torch.manual_seed(0)
    G = triplane.TriPlaneGenerator().eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(2)
        for name, p in G.named_parameters():
            if name.endswith('noise_strength'):
                p.copy_(torch.randn([], generator=g) * 0.1)
    sd = {k: v.detach().clone() for k, v in G.state_dict().items()}
    Gd = G.to(gpu_device)
    z = torch.from_numpy(np.random.RandomState(0).randn(1, 512))
    c = triplane.camera_label(0.5)
    jit = torch.rand(1, 4096, 96, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ws = Gd.mapping(z.to(gpu_device), triplane.conditioning_label(gpu_device))
        out = Gd.synthesis(ws, c=c.to(gpu_device), noise_mode='const', ray_jitter=jit.to(gpu_device), return_dict=True)
    osp = ospec.Spec()
    ws_o = ogen.mapping(sd, osp, z, triplane.conditioning_label(), ops=fast_ops)
    _rel(ws, ws_o, 1e-4, 'ws')
    ref = ogen.synthesis(sd, osp, ws_o, c, jitter=jit, ops=fast_ops)
    _rel(out['planes'][0], ref['planes'][0], 1e-3, 'texture tri-plane')
    _rel(out['planes'][1], ref['planes'][1], 1e-3, 'semantic tri-plane')
    _rel(out['image_raw'], ref['image_raw'], 2e-3, 'raw 64x64 image')
    _rel(out['image'], ref['image'], 2e-3, 'image 512')
    _rel(out['image_seg'], ref['image_seg'], 2e-3, 'seg 512')
    assert out['image'].shape == (1, 3, 512, 512) and out['image_seg'].shape == (1, 19, 512, 512)


def test_extract_shapes_density_cube(golden, gpu_device):
    """extract_shapes.py loop (:99-150) through training.shape_extraction: the device-built lattice is bit-equal to the
    host-built one, chunked and single-launch sigma-only queries agree, and the cube equals the oracle's."""
    from training import shape_extraction as se
    G, cfg, a = _load(golden, gpu_device)
    for N, origin, length in ((12, (0, 0, 0), 1.0), (32, (0.1, -0.2, 0.05), 2.0)):
        dev, _, _ = se.create_samples(N, origin, length, device=gpu_device)
        host, _, _ = se.create_samples(N, origin, length)
        assert torch.equal(dev.cpu(), host)
    N = 12
    img_v, seg_v = t(a['out_img_v'], gpu_device)[:1], t(a['out_seg_v'], gpu_device)[:1]
    before, before_sv = _calls('density_lattice'), _calls('sample_voxel')
    with torch.no_grad():
        cube = se.density_cube(G.synthesis.renderer, img_v, seg_v, voxel_resolution=N, max_batch=500, cube_length=1.0)
        one = se.density_cube(G.synthesis.renderer, img_v, seg_v, voxel_resolution=N, max_batch=None, cube_length=1.0)
        mat = se.density_cube(G.synthesis.renderer, img_v, seg_v, voxel_resolution=N, max_batch=700, cube_length=1.0, materialize=True)
    assert _calls('density_lattice') - before == math.ceil(N ** 3 / 500) + 1          # points generated in the kernel
    assert _calls('sample_voxel') - before_sv == math.ceil(N ** 3 / 700)              # reference-style materialised lattice
    assert cube.shape == (1, N, N, N) and torch.equal(cube, one) and torch.equal(cube, mat)
    sd = {k[len('sd_'):]: t(v) for k, v in a.items() if k.startswith('sd_')}
    samples = 0.9 * oracle_ops.create_samples(N, cube_length=1.0)
    ref = ogen.sample_voxel(sd, ospec.tiny(), img_v.cpu(), seg_v.cpu(), samples, fast_ops)[:, -1]
    _rel(cube.reshape(-1), ref, 2e-4, 'sigma lattice')
    # whole driver (mapping -> tri-planes -> cube) on the GPU vs on the CPU definitions of the same modules
    z, c = t(a['in_z'], gpu_device)[:1], t(a['in_c_cond'], gpu_device)[:1]
    got = se.sample_generator_ide3d(G, None, z, c, max_batch=700, voxel_resolution=N, cube_length=1.0, psi=cfg['truncation_psi'],
                                    noise_mode='const')
    ws = ogen.mapping(sd, ospec.tiny(), z.cpu(), c.cpu(), truncation_psi=cfg['truncation_psi'], ops=fast_ops)
    planes = ogen.backbone(sd, ospec.tiny(), ws, 'const', fast_ops)
    ref = ogen.sample_voxel(sd, ospec.tiny(), planes[0], planes[1], samples, fast_ops)[:, -1]
    _rel(torch.from_numpy(got).reshape(-1), ref, 5e-4, 'driver density cube')


@pytest.mark.parametrize('n,m', [(3, 37), (4, 5), (2, 16), (5, 1), (2, 1000)])
def test_sample_voxel_tiles_that_straddle_images(gpu_device, n, m):
    """sample_voxel works on tiles of 16 rows of the flattened [n * m] point list and the tiles are software-pipelined: when m is
    not a multiple of 16 a tile spans two (m < 16: several) images and its lanes read different tri-planes.  A batched call must equal,
    bit for bit, the per-image calls (one image per call never straddles), for the full rows, the densities and the lattice."""
    from training import triplane
    torch.manual_seed(n * 100 + m)
    R = triplane.TriplaneRenderer(triplane.GeneratorSpec()).to(gpu_device).eval()
    g = torch.Generator().manual_seed(m)
    tex = (torch.randn(n, 96, 64, 64, generator=g) * 0.7).to(gpu_device).contiguous(memory_format=torch.channels_last)
    geo = (torch.randn(n, 96, 64, 64, generator=g) * 0.7).to(gpu_device).contiguous(memory_format=torch.channels_last)
    pts = (torch.rand(n, m, 3, generator=g) * 2.2 - 1.1).to(gpu_device)             # some points outside the planes (zero padding)
    before = _calls('sample_voxel')
    with torch.no_grad():
        full = R.sample_voxel(tex, geo, pts)
        sig = R.sample_voxel(tex, geo, pts, sigma_only=True)
        one = torch.cat([R.sample_voxel(tex[i:i + 1], geo[i:i + 1], pts[i:i + 1]) for i in range(n)])
        lat = R.density_lattice(tex, geo, 16, 2.0 / 15, np.array([-1.0, -1.0, -1.0]), 0.9, 3, m)
        lat1 = torch.cat([R.density_lattice(tex[i:i + 1], geo[i:i + 1], 16, 2.0 / 15, np.array([-1.0, -1.0, -1.0]), 0.9, 3, m) for i in range(n)])
    assert _calls('sample_voxel') - before == 2 + n
    assert full.shape == (n * m, 52) and sig.shape == (n * m,)
    assert torch.equal(full, one), 'batched rows differ from per-image rows'
    assert_close(sig, one[:, -1], rtol=1e-4, atol=1e-5)          # the sigma-only branch sums row 0 of the second layer on the VALU
    assert torch.equal(lat, lat1), 'batched lattice densities differ from per-image ones'


def test_video_sweep_gpu_vs_cpu(golden):
    """gen_videos.py 2x2 grid sweep (training.video_render) on the GPU == the same driver on the CPU definitions."""
    from training import video_render
    gpu = torch.device('cuda:0')
    G, cfg, a = _load(golden, gpu)
    Gc, _, _ = _load(golden, 'cpu')
    kw = dict(w_frames=2, grid_dims=(2, 2), psi=0.7, truncation_cutoff=None, ray_jitter=False)
    fg = list(video_render.gen_interp_frames(G, [3, 5, 7, 11], device=gpu, **kw))
    fc = list(video_render.gen_interp_frames(Gc, [3, 5, 7, 11], device=torch.device('cpu'), **kw))
    assert _calls('frame_u8') >= 2 and _calls('render_rays') >= 2
    for x, y in zip(fg, fc):
        assert x.shape == y.shape == (128, 256, 3)
        diff = (x.cpu().int() - y.int()).abs()
        assert (diff > 1).float().mean() < 5e-3


def test_hipgraph_replay_equals_eager(golden, gpu_device):
    """GraphedRenderer (hipGraph replay of mapping + synthesis) == eager launches, bit for bit, over changing inputs."""
    from training import triplane
    G, cfg, a = _load(golden, gpu_device)
    run = triplane.GraphedRenderer(G, batch=2, device=gpu_device, ray_jitter=False)
    cond = triplane.conditioning_label(gpu_device).repeat(2, 1)
    for seed in (0, 1, 2):
        z = torch.from_numpy(np.random.RandomState(seed).randn(2, G.z_dim)).to(gpu_device).float()
        cam = torch.cat([triplane.camera_label(0.1 * seed - 0.2, device=gpu_device), triplane.camera_label(0.3, device=gpu_device)])
        img_g, seg_g = run(z, cond, cam)
        img_g, seg_g = img_g.clone(), seg_g.clone()
        with torch.no_grad():
            ws = G.mapping(z, cond)
            img_e, seg_e = G.synthesis(ws, c=cam, noise_mode='const', return_seg=True, ray_jitter=False)
        assert torch.equal(img_g, img_e) and torch.equal(seg_g, seg_e)


def test_hybrid_encoder_gpu(golden, gpu_device):
    """HybridEncoder / Encoder forward on the GPU (stride-1 convs on the MFMA kernel, FIR + bias_act on their HIP kernels):
    reference-run fixture (small towers) and the CPU oracle at 256x256 inputs with 512-wide latents."""
    from oracle import encoder as oenc
    from training import encoders
    from test_host_cpu import build_encoder
    for cfg, a in golden('encoder'):
        E = build_encoder(cfg).to(gpu_device)
        with torch.no_grad():
            if cfg['fn'] == 'hybrid_encoder':
                _rel(E(t(a['in_img'], gpu_device), t(a['in_seg'], gpu_device)), a['out_ws'], 1e-3, 'hybrid encoder vs reference')
            else:
                ws, extra = E(t(a['in_x'], gpu_device))
                _rel(ws, a['out_ws'], 1e-3, 'encoder ws vs reference'); _rel(extra, a['out_extra'], 1e-3, 'encoder extra vs reference')
    before = _calls('modconv2d')
    torch.manual_seed(5)
    E = encoders.HybridEncoder(size=256, n_latents_app=10, n_latents_geo=8, w_dim=512).eval()
    sd = {k: v.detach().clone() for k, v in E.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    img = torch.randn(1, 3, 256, 256, generator=g).clamp(-1, 1); seg = torch.randn(1, 19, 256, 256, generator=g)
    with torch.no_grad():
        got = E.to(gpu_device)(img.to(gpu_device), seg.to(gpu_device))
    assert got.shape == (1, 18, 512)
    assert _calls('modconv2d') - before == 2 * (1 + 3 * 6), 'every Conv2dLayer of both towers must run on the HIP conv kernel'
    ref = oenc.hybrid_encoder(sd, img, seg, 10, 8, 512, ops=fast_ops)
    _rel(got, ref, 1e-3, 'hybrid encoder 256 vs oracle')


def test_fp16_blocks_fp32_compute(gpu_device):
    """A spec with StyleGAN2-style fp16 blocks (`num_fp16_res`, conv_clamp 256 — what a released pickle carries,
    inversion/networks.py:1058-1060,1168-1179): the block outputs that cross block boundaries are fp16 (viewer hooks see the
    reference's dtypes), the HIP kernels compute in fp32, the images stay within fp16 storage error of the all-fp32 generator, and
    `force_fp32=True` (viz/renderer.py:439) reproduces the all-fp32 generator bit for bit."""
    from training import triplane
    torch.manual_seed(0)
    spec32 = triplane.tiny_spec(conv_clamp=256)
    spec16 = triplane.tiny_spec(num_fp16_res=2)
    G32 = triplane.TriPlaneGenerator(spec32).eval()
    G16 = triplane.TriPlaneGenerator(spec16).eval()
    G16.load_state_dict(G32.state_dict())
    assert [b.use_fp16 for b in (G16.synthesis.vb8, G16.synthesis.vb16, G16.synthesis.vb32, G16.synthesis.b32, G16.synthesis.b64)] == [False, True, True, True, True]
    assert G16.synthesis.vb32.conv1.conv_clamp == 256
    G32, G16 = G32.to(gpu_device), G16.to(gpu_device)
    z = torch.from_numpy(np.random.RandomState(1).randn(2, G32.z_dim)).to(gpu_device)
    c = torch.cat([triplane.camera_label(0.2), triplane.camera_label(-0.4)]).to(gpu_device)
    cond = triplane.conditioning_label(gpu_device).repeat(2, 1)
    seen = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, n=name: seen.__setitem__(n, o[0].dtype))
             for name, blk in (('vb8', G16.synthesis.vb8), ('vb32', G16.synthesis.vb32), ('b64', G16.synthesis.b64))]
    before = _calls('modconv2d')
    with torch.no_grad():
        ws = G32.mapping(z, cond)
        img32, seg32 = G32.synthesis(ws, c=c, noise_mode='const', ray_jitter=False, return_seg=True)
        img16, seg16 = G16.synthesis(ws, c=c, noise_mode='const', ray_jitter=False, return_seg=True)
        for h in hooks:                  # the forced-fp32 pass below would overwrite the recorded dtypes
            h.remove()
        img16f, seg16f = G16.synthesis(ws, c=c, noise_mode='const', ray_jitter=False, return_seg=True, force_fp32=True)
    assert _calls('modconv2d') - before == 3 * 17, 'every convolution of the fp16 blocks must stay on the HIP kernel (17 launches per pass)'
    assert seen == {'vb8': torch.float32, 'vb32': torch.float16, 'b64': torch.float16}
    assert img16.dtype == torch.float32 and seg16.dtype == torch.float32
    assert torch.equal(img16f, img32) and torch.equal(seg16f, seg32)
    _rel(img16, img32, 5e-3, 'image with fp16 block storage'); _rel(seg16, seg32, 5e-3, 'seg with fp16 block storage')
