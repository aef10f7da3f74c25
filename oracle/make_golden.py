"""Generate tests/golden/*.npz by running the REFERENCE's own Python code (build container only).

    python oracle/make_golden.py            # writes tests/golden/

Every array stored under `out_*` is an output of code imported from /root/reference (its `impl='ref'` op paths,
`training.volumetric_rendering`, `dnnlib.util.sample_from_triplane`, `inversion.networks` blocks, the post-processing
helpers restated from files whose imports cannot be satisfied here are marked so below); `in_*` are the seeded inputs,
`cfg` the JSON call arguments.  The oracle (`oracle/ops.py`, `fast_ops.py`, `generator.py`) is pinned to these files by
tests/test_oracle_golden.py; the HIP path is compared with the oracle (and with these files) by the `-m gpu` tests.
TEST INFRASTRUCTURE ONLY.
"""

import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import  # noqa: E402
from oracle import spec as ospec  # noqa: E402

ref_import.install()

import torch  # noqa: E402
from torch_utils.ops import bias_act as r_bias_act  # noqa: E402  (reference)
from torch_utils.ops import upfirdn2d as r_upfirdn2d  # noqa: E402
from torch_utils.ops import filtered_lrelu as r_flr  # noqa: E402
from training import volumetric_rendering as r_vr  # noqa: E402
from dnnlib import util as r_util  # noqa: E402
import inversion.networks as r_nets  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
LAYER = 'inversion.networks.SynthesisLayer'


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, cases):
    """cases: list of dicts {cfg: {...}, arrays...}; flattened as '<idx>/<key>'."""
    flat = {}
    cfgs = []
    for i, case in enumerate(cases):
        cfgs.append(case.pop('cfg'))
        for k, v in case.items():
            flat[f'{i}/{k}'] = npy(v)
    flat['cfg'] = np.frombuffer(json.dumps(cfgs).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **flat)
    print(f'{name}: {len(cfgs)} cases, {os.path.getsize(path) / 1024:.1f} KiB')


# ---------------------------------------------------------------------------------------------------
def gen_bias_act():
    g = torch.Generator().manual_seed(1)
    cases = []
    variants = [dict(), dict(gain=0.7, clamp=0.9), dict(alpha=0.35, gain=2.0)]
    for act in r_bias_act.activation_funcs:
        for v in variants:
            x = torch.randn(2, 3, 5, 7, generator=g) * 3
            b = torch.randn(3, generator=g)
            y = r_bias_act._bias_act_ref(x, b, dim=1, act=act, **v)
            cases.append(dict(cfg=dict(act=act, dim=1, **v), in_x=x, in_b=b, out_y=y))
    x = torch.randn(4, 6, generator=g); b = torch.randn(4, generator=g)
    cases.append(dict(cfg=dict(act='lrelu', dim=0), in_x=x, in_b=b, out_y=r_bias_act._bias_act_ref(x, b, dim=0, act='lrelu')))
    x = torch.randn(3, 9, generator=g)
    cases.append(dict(cfg=dict(act='relu', dim=1, nobias=True), in_x=x, in_b=torch.zeros(0), out_y=r_bias_act._bias_act_ref(x, None, act='relu')))
    save('bias_act', cases)


UPFIRDN_CASES = [
    # (filter spec, x shape, kwargs)  — the call patterns of SURVEY.md §7 "Recommended tests"
    (dict(f=[1, 3, 3, 1]), (2, 3, 9, 9), dict(up=1, padding=[1, 1, 1, 1], gain=4)),            # after transposed conv
    (dict(f=[1, 3, 3, 1]), (2, 3, 8, 8), dict(up=2, padding=[2, 1, 2, 1], gain=4)),            # skip-image upsample
    (dict(f=[1, 3, 3, 1]), (1, 2, 10, 12), dict(down=2, padding=[1, 1, 1, 1])),                # Conv2dLayer(down=2)
    (dict(f=[1, 3, 3, 1]), (1, 2, 7, 5), dict(up=[2, 1], down=[1, 2], padding=[2, 1, 0, 3], flip_filter=True, gain=1.5)),
    (dict(f=[1, 2, 3, 4, 5, 6, 7, 8, 7, 5, 3, 1], separable=True), (1, 2, 9, 8), dict(up=2, padding=[6, 5, 6, 5], gain=4)),
    (dict(f=[1, 2, 3, 4, 5, 6, 7, 8, 7, 5, 3, 1], separable=True), (1, 2, 20, 18), dict(down=2, padding=[-1, -2, -1, -2], flip_filter=True)),
    (dict(f=[0.5, 1.5, 2.0, 1.0, 0.25], normalize=True), (1, 1, 6, 7), dict(padding=[2, 2, 2, 2])),   # 5x5 non-separable
    (dict(f=None), (1, 2, 4, 4), dict(up=2)),
    (dict(f='rand7x5'), (1, 2, 8, 9), dict(up=3, down=2, padding=[4, 3, 5, 1], gain=0.5)),
    (dict(f='rand23x23'), (1, 1, 6, 6), dict(up=4, padding=[13, 10, 13, 10], gain=16)),         # big 2-D filter (viewer)
    (dict(f=[1, 3, 3, 1]), (1, 2, 6, 6), dict(up=1, padding=[-1, 2, 3, -1])),                   # negative pads / crop
]


def _make_filter(spec, g):
    f = spec.get('f')
    if f == 'rand7x5':
        return torch.randn(7, 5, generator=g)
    if f == 'rand23x23':
        return torch.randn(23, 23, generator=g) * 0.1
    kw = {k: v for k, v in spec.items() if k != 'f'}
    return None if f is None else r_upfirdn2d.setup_filter(f, **kw)


def gen_upfirdn2d():
    g = torch.Generator().manual_seed(2)
    cases = []
    for fspec, shape, kw in UPFIRDN_CASES:
        f = _make_filter(fspec, g)
        x = torch.randn(*shape, generator=g)
        y = r_upfirdn2d._upfirdn2d_ref(x, f, **kw)
        cases.append(dict(cfg=dict(fspec=fspec, **kw), in_x=x, in_f=(f if f is not None else torch.zeros(0)), out_y=y))
    # helper wrappers + setup_filter variants
    f = r_upfirdn2d.setup_filter([1, 3, 3, 1])
    x = torch.randn(1, 2, 6, 6, generator=g)
    cases.append(dict(cfg=dict(helper='upsample2d'), in_x=x, in_f=f, out_y=r_upfirdn2d.upsample2d(x, f, impl='ref')))
    cases.append(dict(cfg=dict(helper='downsample2d'), in_x=x, in_f=f, out_y=r_upfirdn2d.downsample2d(x, f, impl='ref')))
    cases.append(dict(cfg=dict(helper='filter2d'), in_x=x, in_f=f, out_y=r_upfirdn2d.filter2d(x, f, impl='ref')))
    for kw in (dict(f=[1, 3, 3, 1]), dict(f=[1, 2, 1], gain=4), dict(f=list(range(1, 10))), dict(f=[1, 2, 3], flip_filter=True, normalize=False),
               dict(f=[[1, 2], [3, 4]], gain=2), dict(f=3.0)):
        cases.append(dict(cfg=dict(setup_filter=kw), out_y=r_upfirdn2d.setup_filter(**kw)))
    save('upfirdn2d', cases)


def gen_filtered_lrelu():
    g = torch.Generator().manual_seed(3)
    cases = []
    f12 = r_upfirdn2d.setup_filter([1, 2, 3, 4, 5, 6, 6, 5, 4, 3, 2, 1], separable=True)
    f8 = r_upfirdn2d.setup_filter([1, 3, 5, 7, 7, 5, 3, 1], separable=True)
    f4 = r_upfirdn2d.setup_filter([1, 3, 3, 1])
    specs = [
        dict(fu=f12, fd=f12, up=2, down=2, padding=[9, 10, 9, 10], clamp=1.2, shape=(2, 3, 10, 10)),
        dict(fu=f8, fd=f12, up=4, down=2, padding=[8, 9, 8, 9], gain=1.1, slope=0.3, shape=(1, 2, 9, 7)),
        dict(fu=f4, fd=f4, up=2, down=1, padding=[2, 1, 2, 1], flip_filter=True, shape=(1, 2, 6, 6)),
        dict(fu=None, fd=None, up=1, down=1, padding=0, clamp=0.5, shape=(1, 3, 5, 5)),
        dict(fu=f12, fd=None, up=2, down=1, padding=[6, 5, 6, 5], shape=(1, 1, 8, 8)),
    ]
    # round 6: more reference-run shapes — the compile-time (up, down, taps) instances of csrc/filtered_lrelu.hip over several
    # 32 x 32 tiles, ASYMMETRIC taps (a flip error shows), a non-separable 5 x 5 pair, every scalar option at once, odd sizes and paddings
    a12 = r_upfirdn2d.setup_filter([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12], separable=True)
    a24 = r_upfirdn2d.setup_filter(list(range(1, 25)), separable=True)
    a8 = r_upfirdn2d.setup_filter([1, 2, 3, 4, 5, 6, 7, 8], separable=True)
    n5 = r_upfirdn2d.setup_filter([1, 2, 3, 2, 1], separable=False)
    n5[0, 1] *= 3; n5 = n5 / n5.sum()                       # neither symmetric nor an outer product
    specs += [
        dict(fu=a12, fd=a12, up=2, down=2, padding=[10, 11, 10, 11], flip_filter=True, clamp=0.8, gain=1.3, slope=0.1, shape=(1, 2, 40, 40)),
        dict(fu=a12, fd=a12, up=2, down=2, padding=[10, 11, 10, 11], shape=(1, 1, 37, 21)),
        dict(fu=a12, fd=a12, up=4, down=2, padding=[11, 10, 12, 9], slope=0.25, shape=(1, 2, 12, 11)),
        dict(fu=a24, fd=a12, up=4, down=2, padding=[17, 16, 17, 16], clamp=2.0, shape=(1, 1, 10, 13)),
        dict(fu=a8, fd=a8, up=2, down=2, padding=[6, 7, 6, 7], gain=0.7, shape=(2, 2, 9, 9)),
        dict(fu=None, fd=a12, up=1, down=2, padding=[5, 6, 5, 6], shape=(1, 2, 20, 18)),
        dict(fu=a12, fd=a12, up=2, down=4, padding=[10, 11, 10, 11], shape=(1, 2, 16, 16)),
        dict(fu=n5, fd=n5, up=2, down=2, padding=[3, 4, 4, 3], flip_filter=True, clamp=1.0, shape=(1, 2, 8, 9)),
    ]
    for s in specs:
        shape = s.pop('shape')
        x = torch.randn(*shape, generator=g)
        b = torch.randn(shape[1], generator=g)
        y = r_flr._filtered_lrelu_ref(x, b=b, **s)
        cfg = {k: v for k, v in s.items() if k not in ('fu', 'fd')}
        cases.append(dict(cfg=cfg, in_x=x, in_b=b, in_fu=(s['fu'] if s['fu'] is not None else torch.zeros(0)),
                          in_fd=(s['fd'] if s['fd'] is not None else torch.zeros(0)), out_y=y))
    save('filtered_lrelu', cases)


def gen_volumetric():
    cases = []
    dev = 'cpu'
    # rays
    for n, steps, fov, res, t0, t1 in ((2, 5, 18, (4, 4), 2.25, 3.3), (1, 7, 30, (6, 3), 0.5, 1.5)):
        pts, z, d = r_vr.get_initial_rays_trig(n, steps, dev, fov, res, t0, t1)
        cases.append(dict(cfg=dict(fn='get_initial_rays_trig', n=n, num_steps=steps, fov=fov, resolution=list(res), ray_start=t0, ray_end=t1),
                          out_points=pts, out_z=z, out_d=d))
    # cameras
    for yaw in (-0.5, 0.0, 0.5):
        cam, phi, theta = r_vr.sample_camera_positions(dev, n=1, r=2.7, horizontal_mean=yaw + math.pi / 2, vertical_mean=math.pi / 2, mode=None)
        c2w = r_vr.create_cam2world_matrix(-cam, cam, device=dev)
        cases.append(dict(cfg=dict(fn='gen_images_pose', yaw=yaw), out_cam=cam, out_c2w=c2w))
    for t in (0.0, 0.3, 0.77):
        h = math.pi / 2 - 0.5 * math.sin(2 * math.pi * t)
        v = math.pi / 2 - 0.05 + 0.25 * math.cos(2 * math.pi * t)
        c2w = r_vr.LookAtPoseSampler.sample(h, v, torch.tensor([0, 0, 0.2]), radius=2.7, device=dev)
        cases.append(dict(cfg=dict(fn='lookat', h=h, v=v, lookat=[0, 0, 0.2], radius=2.7), out_c2w=c2w))
    # jitter + transform
    pts, z, d = r_vr.get_initial_rays_trig(2, 6, dev, 18, (4, 4), 2.25, 3.3)
    cam, _, _ = r_vr.sample_camera_positions(dev, n=2, r=2.7, horizontal_mean=0.4 + math.pi / 2, vertical_mean=1.4, mode=None)
    c2w = r_vr.create_cam2world_matrix(-cam, cam, device=dev)
    torch.manual_seed(11); jitter = torch.rand(z.shape)
    torch.manual_seed(11)
    tp, tz, td, to, _, _ = r_vr.transform_sampled_points(pts, z, d, dev, h_stddev=0, v_stddev=0, camera=c2w, mode=None)
    cases.append(dict(cfg=dict(fn='transform_sampled_points', n=2, num_steps=6, fov=18, resolution=[4, 4], ray_start=2.25, ray_end=3.3),
                      in_c2w=c2w, in_jitter=jitter, out_points=tp, out_z=tz, out_dirs=td, out_origins=to))
    # compositing
    g = torch.Generator().manual_seed(5)
    rs = torch.randn(2, 9, 13, 7, generator=g)
    rs[..., -1] *= 4
    zz = torch.sort(torch.rand(2, 9, 13, 1, generator=g) * 1.05 + 2.25, dim=2)[0]
    dd = torch.randn(2, 9, 3, generator=g)
    variants = [dict(clamp_mode='softplus'), dict(clamp_mode='relu'), dict(clamp_mode='softplus', last_back=True),
                dict(clamp_mode='softplus', white_back=True, max_depth=5.0), dict(clamp_mode='relu', fill_mode='weight')]
    for v in variants:
        rgb, depth, w = r_vr.fancy_integration(rs.clone(), dd, zz, dev, noise_std=0, **v)
        cases.append(dict(cfg=dict(fn='fancy_integration', **v), in_rs=rs, in_z=zz, in_d=dd, out_rgb=rgb, out_depth=depth, out_w=w))
    rs3 = rs[..., [0, 1, 2, 6]].clone()
    rgb, depth, w = r_vr.fancy_integration(rs3.clone(), dd, zz, dev, noise_std=0, clamp_mode='relu', fill_mode='debug')
    cases.append(dict(cfg=dict(fn='fancy_integration', clamp_mode='relu', fill_mode='debug'), in_rs=rs3, in_z=zz, in_d=dd, out_rgb=rgb, out_depth=depth, out_w=w))
    torch.manual_seed(21); noise = torch.randn(2, 9, 13, 1)
    torch.manual_seed(21)
    rgb, depth, w = r_vr.fancy_integration(rs.clone(), dd, zz, dev, noise_std=0.5, clamp_mode='softplus')
    cases.append(dict(cfg=dict(fn='fancy_integration', clamp_mode='softplus', noise_std=0.5), in_rs=rs, in_z=zz, in_d=dd, in_noise=noise,
                      out_rgb=rgb, out_depth=depth, out_w=w))
    # importance sampling
    zmid = 0.5 * (zz[:, :, :-1, 0] + zz[:, :, 1:, 0]).reshape(18, 12)
    wts = torch.rand(18, 11, generator=g)
    cases.append(dict(cfg=dict(fn='sample_pdf', N_importance=9), in_bins=zmid, in_w=wts, out_samples=r_vr.sample_pdf(zmid, wts, 9, det=True)))
    # ... with random draws (det=False): the reference's first RNG call after seeding is `torch.rand(rays, N_importance)`,
    # so the draws can be stored next to the result.  Includes empty bins (denom < eps) and an all-zero ray.
    g2 = torch.Generator().manual_seed(33)
    bins2 = torch.sort(torch.rand(40, 23, generator=g2) * 1.05 + 2.25, dim=1)[0]
    w2 = torch.rand(40, 22, generator=g2)
    w2[:, 5:9] = 0
    w2[3] = 0
    w2[7, :20] = 0
    torch.manual_seed(77); u2 = torch.rand(40, 24)
    torch.manual_seed(77); s2 = r_vr.sample_pdf(bins2, w2, 24, det=False)
    cases.append(dict(cfg=dict(fn='sample_pdf_rand', N_importance=24), in_bins=bins2, in_w=w2, in_u=u2, out_samples=s2))
    save('volumetric', cases)


def gen_triplane():
    g = torch.Generator().manual_seed(6)
    cases = []
    for (B, C, H, M, spread) in ((2, 4, 8, 40, 1.3), (1, 8, 16, 64, 0.8), (1, 3, 5, 30, 1.1)):
        grid = torch.randn(B, 3 * C, H, H, generator=g)
        co = (torch.rand(B, M, 3, generator=g) * 2 - 1) * spread
        co[:, 0] = torch.tensor([1.0, -1.0, 0.0])      # corners / edges
        co[:, 1] = torch.tensor([-1.0 + 1.0 / H, 1.0 - 1.0 / H, 0.999999])
        out = r_util.sample_from_triplane(co, grid)
        cases.append(dict(cfg=dict(B=B, C=C, H=H, M=M), in_grid=grid, in_coords=co, out_feat=out))
    # round 6: the generator's own channel count (32 per plane: the kernels' compiled width), a frustum-shaped ray grid the LDS-staged
    # kernel tiles (8 x 8 rays x 4 steps), corner and far-outside coordinates (zeros padding, never clamped)
    for (B, C, H, rays, steps, spread) in ((2, 32, 16, 8, 4, 0.55), (1, 32, 32, 16, 8, 1.6)):
        grid = torch.randn(B, 3 * C, H, H, generator=g)
        org = torch.tensor([0.05, -0.1, 2.7]) * spread / 1.6
        px = torch.stack(torch.meshgrid(torch.linspace(-0.3, 0.3, rays), torch.linspace(0.3, -0.3, rays), indexing='xy'), -1).reshape(-1, 2)
        d = torch.nn.functional.normalize(torch.cat([px, -torch.ones(rays * rays, 1) * 1.9], 1), dim=1)
        t = torch.linspace(1.2, 2.9, steps) * spread
        co = (org + d[:, None, :] * t[None, :, None]).reshape(1, -1, 3).repeat(B, 1, 1)
        co = co + torch.randn(co.shape, generator=g) * 1e-3
        co[:, 3] = torch.tensor([1.0, -1.0, 1.0]); co[:, 5] = torch.tensor([-3.0, 0.25, 1e6]); co[:, 7] = torch.tensor([37.0, -1e9, 0.5])      # (finite only: ATen's CPU and
        # CUDA kernels disagree on non-finite coordinates — NaN vs 0 — so the reference has no answer to pin there)
        out = r_util.sample_from_triplane(co, grid)
        cases.append(dict(cfg=dict(B=B, C=C, H=H, M=co.shape[1], ray_grid=[rays, rays, steps]), in_grid=grid, in_coords=co, out_feat=out))
    save('triplane', cases)


def _randomize_zero_params(module, g):
    """Give zero-initialised biases / noise strengths non-trivial values so that they are exercised."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith('noise_strength'):
                p.copy_(torch.randn([], generator=g) * 0.3)
            elif name.endswith('bias') and float(p.abs().sum()) == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)


def gen_networks():
    g = torch.Generator().manual_seed(7)
    torch.manual_seed(7)
    cases = []
    # modulated_conv2d
    f = r_upfirdn2d.setup_filter([1, 3, 3, 1])
    for kw in (dict(up=1, padding=1, demodulate=True, fused_modconv=True, flip_weight=True),
               dict(up=1, padding=1, demodulate=True, fused_modconv=False, flip_weight=True),
               dict(up=2, padding=1, demodulate=True, fused_modconv=True, flip_weight=False),
               dict(up=1, padding=0, demodulate=False, fused_modconv=True, flip_weight=True, k=1)):
        k = kw.pop('k', 3)
        x = torch.randn(2, 5, 6, 6, generator=g)
        w = torch.randn(4, 5, k, k, generator=g)
        s = torch.randn(2, 5, generator=g) + 1
        nz = torch.randn(kw['up'] * 6, kw['up'] * 6, generator=g) * 0.1
        y = r_nets.modulated_conv2d(x=x, weight=w, styles=s, noise=nz, resample_filter=(f if kw['up'] > 1 else None), **kw)
        cases.append(dict(cfg=dict(fn='modulated_conv2d', **kw), in_x=x, in_w=w, in_s=s, in_noise=nz, out_y=y))
    # mapping network
    m = r_nets.MappingNetwork(z_dim=16, c_dim=25, w_dim=12, num_ws=5, num_layers=3).eval()
    _randomize_zero_params(m, g)
    with torch.no_grad():
        m.w_avg.copy_(torch.randn(12, generator=g) * 0.1)
        z = torch.randn(3, 16, generator=g); c = torch.randn(3, 25, generator=g)
        case = dict(cfg=dict(fn='mapping', z_dim=16, c_dim=25, w_dim=12, num_ws=5, num_layers=3), in_z=z, in_c=c,
                    out_ws=m(z, c), out_ws_trunc=m(z, c, truncation_psi=0.6, truncation_cutoff=3))
    for k, v in m.state_dict().items():
        case['sd_mapping.' + k] = v
    cases.append(case)
    save('networks', cases)


class _Box(torch.nn.Module):
    pass


def build_reference_generator(sp, seed):
    """The generator topology of training/triplane.py assembled from the REFERENCE's modules."""
    torch.manual_seed(seed)
    G = _Box(); G.synthesis = _Box(); G.synthesis.renderer = _Box(); G.synthesis.renderer.decoder = _Box()
    nws = (1 + 2 * (len(sp.voxel_resolutions()) - 1)) + 2 * len(sp.sr_resolutions()) + 1
    G.mapping = r_nets.MappingNetwork(z_dim=sp.z_dim, c_dim=sp.c_dim, w_dim=sp.w_dim, num_ws=nws, num_layers=sp.mapping_layers)
    pc = 3 * sp.plane_channels
    for res in sp.voxel_resolutions():
        cin = sp.voxel_width(res // 2) if res > 4 else 0
        setattr(G.synthesis, f'vb{res}', r_nets.SegSynthesisBlock(cin, sp.voxel_width(res), w_dim=sp.w_dim, resolution=res, img_channels=pc,
                                                                  seg_channels=pc, is_last=False, conv_clamp=sp.conv_clamp, layer_name=LAYER))
    d = G.synthesis.renderer.decoder
    d.geo0 = r_nets.FullyConnectedLayer(sp.plane_channels, sp.decoder_hidden, activation='softplus')
    d.geo1 = r_nets.FullyConnectedLayer(sp.decoder_hidden, 1 + sp.seg_channels)
    d.tex0 = r_nets.FullyConnectedLayer(sp.plane_channels, sp.decoder_hidden, activation='softplus')
    d.tex1 = r_nets.FullyConnectedLayer(sp.decoder_hidden, sp.feature_channels)
    cin = sp.feature_channels
    widths = sp.sr_widths()
    for res in sp.sr_resolutions():
        setattr(G.synthesis, f'b{res}', r_nets.SegSynthesisBlock(cin, widths[res], w_dim=sp.w_dim, resolution=res, img_channels=sp.img_channels,
                                                                 seg_channels=sp.seg_channels, is_last=(res == sp.img_resolution),
                                                                 conv_clamp=sp.conv_clamp, layer_name=LAYER))
        cin = widths[res]
    g = torch.Generator().manual_seed(seed + 1)
    _randomize_zero_params(G, g)
    with torch.no_grad():
        G.mapping.w_avg.copy_(torch.randn(sp.w_dim, generator=g) * 0.1)
    return G.eval()


def reference_synthesis(G, sp, ws, c, jitter_seed, noise_mode='const'):
    """G.synthesis of SURVEY.md §3.5 evaluated with reference functions only."""
    syn = G.synthesis
    ws = ws.to(torch.float32)
    voxel_ws, block_ws, w_idx = [], [], 0
    for res in sp.voxel_resolutions():
        blk = getattr(syn, f'vb{res}')
        voxel_ws.append(ws.narrow(1, w_idx, blk.num_conv + blk.num_torgb)); w_idx += blk.num_conv
    for res in sp.sr_resolutions():
        blk = getattr(syn, f'b{res}')
        block_ws.append(ws.narrow(1, w_idx, blk.num_conv + blk.num_torgb)); w_idx += blk.num_conv
    x = img_v = seg_v = None
    for res, cur in zip(sp.voxel_resolutions(), voxel_ws):
        x, img_v, seg_v = getattr(syn, f'vb{res}')(x, img_v, seg_v, cur, noise_mode=noise_mode)
    n = ws.shape[0]
    size = sp.render_size
    pts, z, d = r_vr.get_initial_rays_trig(n, sp.num_steps, 'cpu', sp.fov, (size, size), sp.ray_start, sp.ray_end)
    cam2world = c[:, :16].reshape(-1, 4, 4)
    torch.manual_seed(jitter_seed); jitter = torch.rand(z.shape)
    torch.manual_seed(jitter_seed)
    wp, z, _, _, _, _ = r_vr.transform_sampled_points(pts, z, d, 'cpu', h_stddev=0, v_stddev=0, camera=cam2world, mode=None)
    flat = wp.reshape(n, -1, 3)
    dec = syn.renderer.decoder
    gfeat = dec.geo1(dec.geo0(r_util.sample_from_triplane(flat, seg_v)))
    tfeat = dec.tex1(dec.tex0(r_util.sample_from_triplane(flat, img_v)))
    out = torch.cat([tfeat, gfeat[:, 1:], gfeat[:, :1]], 1).reshape(n, size * size, sp.num_steps, -1)
    feat, depth, weights = r_vr.fancy_integration(out, d, z, 'cpu', noise_std=0, clamp_mode=sp.clamp_mode)
    feat = feat.permute(0, 2, 1).reshape(n, -1, size, size)
    depth = depth.permute(0, 2, 1).reshape(n, 1, size, size)
    fc = sp.feature_channels
    s2 = sp.sr_resolutions()[0] // 2
    up = lambda t: torch.nn.functional.interpolate(t, size=(s2, s2), mode='bilinear', align_corners=False)
    x, img, seg = up(feat[:, :fc]), up(feat[:, :sp.img_channels]), up(feat[:, fc:])
    for res, cur in zip(sp.sr_resolutions(), block_ws):
        x, img, seg = getattr(syn, f'b{res}')(x, img, seg, cur, noise_mode=noise_mode)
    return dict(img=img, seg=seg, feat=feat, depth=depth, img_v=img_v, seg_v=seg_v, jitter=jitter[..., 0],
                sample_out=out[:, :4].reshape(-1, out.shape[-1]), sample_pts=flat[:, :4 * sp.num_steps])


def gen_generator():
    sp = ospec.tiny()
    G = build_reference_generator(sp, seed=100)
    g = torch.Generator().manual_seed(101)
    z = torch.from_numpy(np.random.RandomState(3).randn(2, sp.z_dim))
    cs = torch.tensor([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 2.7, 0, 0, 0, 1, 4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).float().reshape(1, -1).repeat(2, 1)
    cams = []
    for yaw in (-0.5, 0.35):
        cam, _, _ = r_vr.sample_camera_positions('cpu', n=1, r=2.7, horizontal_mean=yaw + math.pi / 2, vertical_mean=math.pi / 2, mode=None)
        c2w = r_vr.create_cam2world_matrix(-cam, cam, device='cpu').reshape(1, -1)
        cams.append(torch.cat((c2w, torch.tensor([4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).reshape(1, -1)), -1))
    c = torch.cat(cams, 0)
    with torch.no_grad():
        ws = G.mapping(z, cs, truncation_psi=0.7)
        res = reference_synthesis(G, sp, ws, c, jitter_seed=55)
    case = dict(cfg=dict(spec='tiny', jitter_seed=55, truncation_psi=0.7), in_z=z, in_c_cond=cs, in_c=c, out_ws=ws)
    for k, v in res.items():
        case[('in_' if k in ('jitter', 'sample_pts') else 'out_') + k] = v
    for k, v in G.state_dict().items():
        case['sd_' + k] = v
    save('generator_tiny', [case])


def _sd_fingerprint(module):
    """Order-sensitive checksum of a state dict (float64): sum(|t|) and sum(t * ramp) per tensor."""
    out = []
    for k, v in module.state_dict().items():
        v = v.double().flatten()
        out.append([float(v.abs().sum()), float((v * torch.linspace(0.5, 1.5, v.numel(), dtype=torch.float64)).sum())])
    return torch.tensor(out, dtype=torch.float64)


def gen_encoder():
    """HybridEncoder / Encoder forward (inversion/networks.py:1559-1665).  The weights (tens of MB) are not stored: both
    sides are built under the same torch seed (parameters are created in the same order), the fixture keeps a
    fingerprint of the reference state dict so the test can prove the weights are the same before comparing outputs."""
    cases = []
    g = torch.Generator().manual_seed(17)
    torch.manual_seed(1234)
    E = r_nets.HybridEncoder(size=16, n_latents_app=3, n_latents_geo=2, w_dim=8, add_dim=0, input_img_dim=3, input_seg_dim=19).eval()
    img = torch.randn(2, 3, 16, 16, generator=g); seg = torch.randn(2, 19, 16, 16, generator=g)
    with torch.no_grad():
        out = E(img, seg)
    cases.append(dict(cfg=dict(fn='hybrid_encoder', seed=1234, size=16, n_latents_app=3, n_latents_geo=2, w_dim=8, add_dim=0,
                               keys=list(E.state_dict().keys())),
                      in_img=img, in_seg=seg, out_ws=out, sd_fingerprint=_sd_fingerprint(E)))
    torch.manual_seed(4321)
    E1 = r_nets.Encoder(size=8, n_latents=2, w_dim=4, add_dim=2, input_dim=3).eval()
    x = torch.randn(3, 3, 8, 8, generator=g)
    with torch.no_grad():
        ws, extra = E1(x)
    cases.append(dict(cfg=dict(fn='encoder', seed=4321, size=8, n_latents=2, w_dim=4, add_dim=2, keys=list(E1.state_dict().keys())),
                      in_x=x, out_ws=ws, out_extra=extra, sd_fingerprint=_sd_fingerprint(E1)))
    save('encoder', cases)


def gen_bisenet():
    """`BiSeNet(n_classes=20)` forward (inversion/BiSeNet.py:229-256) in eval mode on a synthetic state dict that is a function of the parameter
    NAMES (oracle/face_parsing.py `synthetic_state_dict`: the 53 MB of weights are not stored), plus the label pipeline of
    dnnlib/seg_tools.py:100-123 restated on the logits (that module imports torchvision, which is not installed: `id_remap` / `scatter` are the
    two-line bodies executed from the reference source text, as for gen_post)."""
    import re
    from inversion.BiSeNet import BiSeNet
    from oracle import face_parsing as ofp
    torch.manual_seed(5)
    net = BiSeNet(n_classes=20).eval()
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(ofp.synthetic_state_dict(shapes))
    g = torch.Generator().manual_seed(23)
    cases = []
    for n, h, w in ((2, 64, 96), (1, 96, 64)):
        x = torch.randn(n, 3, h, w, generator=g)
        with torch.no_grad():
            out = net(x)[0]
        cases.append(dict(cfg=dict(fn='bisenet', n_classes=20, shapes=shapes), in_x=x, out_logits=out))
    ns = dict(torch=torch, F=torch.nn.functional)
    seg_src = open(os.path.join(ref_import.REFERENCE_ROOT, 'dnnlib', 'seg_tools.py')).read()
    exec(re.search(r'remap_list = .*?\n', seg_src).group(0) + re.search(r'def id_remap\(.*?\n\n', seg_src, re.S).group(0)
         + re.search(r'def scatter\(.*?return input_label\.scatter_\(1, condition_img\.long\(\), 1\)\n', seg_src, re.S).group(0), ns)
    logits = cases[1]['out_logits']
    seg = ns['id_remap'](logits.argmax(1, keepdim=True), 'celebahq')
    cases.append(dict(cfg=dict(fn='labels', case=1), out_remap=seg.to(torch.uint8), out_onehot=ns['scatter'](seg, label_size=(96, 64)).to(torch.uint8)))
    save('bisenet', cases)


def gen_post():
    """mask2color / layout_grid / create_samples.  dnnlib/seg_tools.py and extract_shapes.py import packages that are
    not installed (torchvision, BiSeNet, mrcfile), so their few-line function bodies are executed here from the
    reference SOURCE TEXT (functions extracted by name) instead of by module import."""
    import re
    ns = dict(torch=torch, np=np)
    seg_src = open(os.path.join(ref_import.REFERENCE_ROOT, 'dnnlib', 'seg_tools.py')).read()
    cm = re.search(r'COLOR_MAP = \{.*?\}\n', seg_src, re.S).group(0)
    fn = re.search(r'def mask2color\(masks\):.*?return sample_mask\n', seg_src, re.S).group(0)
    exec(cm + fn, ns)
    ex_src = open(os.path.join(ref_import.REFERENCE_ROOT, 'extract_shapes.py')).read()
    cs = re.search(r'def create_samples\(.*?return samples\.unsqueeze\(0\), voxel_origin, voxel_size\n', ex_src, re.S).group(0)
    exec(cs, ns)
    g = torch.Generator().manual_seed(9)
    img = torch.randn(2, 3, 8, 12, generator=g) * 0.8
    seg = torch.randn(2, 19, 8, 12, generator=g)
    seg[0, 3, 0, 0] = seg[0, 7, 0, 0] = 9.0                                   # tie -> first index
    col = ns['mask2color'](seg)
    frame = torch.cat([img, (col / 255. - 0.5) / 0.5], dim=-1)
    u8 = r_util.layout_grid(frame, grid_w=2, grid_h=1, to_numpy=False)        # [H, 2*(2W), 3]
    samples, _, _ = ns['create_samples'](N=6, voxel_origin=[0, 0, 0], cube_length=1.0)
    samples2, origin2, size2 = ns['create_samples'](N=16, voxel_origin=[0.1, -0.2, 0.05], cube_length=2.0)
    cases = [dict(cfg=dict(fn='frame'), in_img=img, in_seg=seg, out_color=col, out_grid_u8=u8),
             dict(cfg=dict(fn='create_samples', N=6, cube_length=1.0), out_samples=samples),
             dict(cfg=dict(fn='create_samples', N=16, cube_length=2.0, voxel_origin=[0.1, -0.2, 0.05], voxel_size=float(size2)),
                  out_samples=samples2, out_origin=torch.from_numpy(np.asarray(origin2, dtype=np.float64)))]
    save('post', cases)


if __name__ == '__main__':
    torch.set_num_threads(4)
    if len(sys.argv) > 1:                       # regenerate only the named fixtures: python oracle/make_golden.py post ...
        for name in sys.argv[1:]:
            globals()['gen_' + name]()
        sys.exit(0)
    gen_bias_act()
    gen_upfirdn2d()
    gen_filtered_lrelu()
    gen_volumetric()
    gen_triplane()
    gen_networks()
    gen_generator()
    gen_encoder()
    gen_bisenet()
    gen_post()
