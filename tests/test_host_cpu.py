"""Host-side logic of the product on CPU (no GPU needed): the reference call surfaces dispatch to their PyTorch
definitions for CPU tensors and must reproduce the reference-generated golden vectors; the C-ABI library loads and
exports every symbol the header declares; a missing library is a hard error."""

import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

from oracle import fast_ops
from oracle import generator as ogen
from oracle import ops as oracle_ops
from oracle import spec as ospec
from util import assert_close, filter_from, t

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- boundary ------------------------------------------------------------------------------------------

def _header_symbols():
    src = open(os.path.join(ROOT, 'include', 'ide3d_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ide3d_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from torch_utils import hip_plugin
    path = hip_plugin.lib_path()
    assert os.path.isfile(path), f'{path} missing: run __graft_entry__.build()'
    lib = ctypes.CDLL(path)
    declared = _header_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/ide3d_hip.h but not exported'
    assert set(hip_plugin.EXPORTED_SYMBOLS) == set(declared)
    lib.ide3d_abi_version.restype = ctypes.c_int
    lib.ide3d_build_arch.restype = ctypes.c_char_p
    assert lib.ide3d_abi_version() == hip_plugin._ABI_VERSION == 8
    lib.ide3d_build_flags.restype = ctypes.c_char_p
    assert lib.ide3d_build_flags() == b'', 'the shipped library is a release build: no experiment knobs'
    assert lib.ide3d_build_arch() == b'gfx950'


def test_ctypes_structs_match_header_layout():
    """Field order of the ctypes mirrors == field order of the C structs."""
    from torch_utils import hip_plugin
    src = open(os.path.join(ROOT, 'include', 'ide3d_hip.h')).read()
    for cname, cls in (('ide3d_upfirdn2d_params', hip_plugin._UpfirdnParams), ('ide3d_filtered_lrelu_params', hip_plugin._FlreluParams),
                       ('ide3d_render_params', hip_plugin._RenderParams), ('ide3d_modconv_params', hip_plugin._ModconvParams),
                       ('ide3d_lattice', hip_plugin._Lattice), ('ide3d_style_job', hip_plugin._StyleJob), ('ide3d_fold_job', hip_plugin._FoldJob),
                       ('ide3d_modconv_plan_info', hip_plugin._ModconvPlanInfo)):
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (cname, cname), src, re.S).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        names = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(','):
                m = re.search(r'([A-Za-z_][A-Za-z0-9_]*)\s*(\[\d+\])?\s*$', part.strip())
                names.append(m.group(1))
        assert names == [f[0] for f in cls._fields_], cname


def test_missing_library_is_a_hard_error(monkeypatch):
    from torch_utils import hip_plugin
    monkeypatch.setenv('IDE3D_HIP_LIB', '/nonexistent/libide3d_hip.so')
    monkeypatch.setattr(hip_plugin, '_lib', None)
    with pytest.raises(RuntimeError, match='not found'):
        hip_plugin.load()


def test_product_does_not_import_oracle():
    """The product tree must never reference the oracle."""
    overlay = os.path.join(ROOT, 'ide-3d_amd')
    for dirpath, _dirs, files in os.walk(overlay):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dirpath, fn)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, os.path.join(dirpath, fn)


# ---- op surfaces on CPU tensors -----------------------------------------------------------------------------

def test_bias_act_cpu(golden):
    from torch_utils.ops import bias_act
    assert list(bias_act.activation_funcs) == ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']
    assert [s.cuda_idx for s in bias_act.activation_funcs.values()] == list(range(1, 10))
    for cfg, a in golden('bias_act'):
        kw = {k: cfg[k] for k in ('alpha', 'gain', 'clamp') if k in cfg}
        b = None if cfg.get('nobias') else t(a['in_b'])
        y = bias_act.bias_act(t(a['in_x']), b, dim=cfg['dim'], act=cfg['act'], **kw)
        assert_close(y, a['out_y'], rtol=1e-6, atol=1e-6, what=str(cfg))


def test_upfirdn2d_cpu(golden):
    from torch_utils.ops import upfirdn2d
    for cfg, a in golden('upfirdn2d'):
        if 'fspec' in cfg:
            kw = {k: v for k, v in cfg.items() if k != 'fspec'}
            y = upfirdn2d.upfirdn2d(t(a['in_x']), filter_from(a['in_f']), **kw)
            assert_close(y, a['out_y'], rtol=1e-5, atol=1e-5, what=str(cfg))
        elif 'helper' in cfg:
            y = getattr(upfirdn2d, cfg['helper'])(t(a['in_x']), t(a['in_f']))
            assert_close(y, a['out_y'], rtol=1e-5, atol=1e-6, what=cfg['helper'])
        else:
            assert_close(upfirdn2d.setup_filter(**cfg['setup_filter']), a['out_y'], rtol=1e-6, atol=1e-7, what=str(cfg))


def test_filtered_lrelu_cpu(golden):
    from torch_utils.ops import filtered_lrelu
    for cfg, a in golden('filtered_lrelu'):
        y = filtered_lrelu.filtered_lrelu(t(a['in_x']), fu=filter_from(a['in_fu']), fd=filter_from(a['in_fd']), b=t(a['in_b']), **cfg)
        assert_close(y, a['out_y'], rtol=1e-5, atol=1e-5, what=str(cfg))


def test_volumetric_cpu(golden):
    from training import volumetric_rendering as vr
    for cfg, a in golden('volumetric'):
        fn = cfg['fn']
        if fn == 'get_initial_rays_trig':
            p, z, d = vr.get_initial_rays_trig(cfg['n'], cfg['num_steps'], 'cpu', cfg['fov'], tuple(cfg['resolution']), cfg['ray_start'], cfg['ray_end'])
            assert_close(p, a['out_points'], rtol=0, atol=0, what='points')
            assert_close(z, a['out_z'], rtol=0, atol=0, what='z')
            assert_close(d, a['out_d'], rtol=0, atol=0, what='dirs')
        elif fn == 'gen_images_pose':
            cam, _, _ = vr.sample_camera_positions('cpu', n=1, r=2.7, horizontal_mean=cfg['yaw'] + math.pi / 2, vertical_mean=math.pi / 2, mode=None)
            assert_close(vr.create_cam2world_matrix(-cam, cam, device='cpu'), a['out_c2w'], rtol=0, atol=1e-7, what='c2w')
        elif fn == 'lookat':
            c2w = vr.LookAtPoseSampler.sample(cfg['h'], cfg['v'], torch.tensor(cfg['lookat']), radius=cfg['radius'])
            assert_close(c2w, a['out_c2w'], rtol=0, atol=1e-6, what='lookat')
        elif fn == 'transform_sampled_points':
            p, z, d = vr.get_initial_rays_trig(cfg['n'], cfg['num_steps'], 'cpu', cfg['fov'], tuple(cfg['resolution']), cfg['ray_start'], cfg['ray_end'])
            tp, tz, td, to, _, _ = vr.transform_sampled_points(p, z, d, 'cpu', h_stddev=0, v_stddev=0, camera=t(a['in_c2w']), mode=None,
                                                               jitter=t(a['in_jitter']))
            assert_close(tp, a['out_points'], rtol=1e-6, atol=1e-6, what='world points')
            assert_close(tz, a['out_z'], rtol=0, atol=0, what='z')
            assert_close(td, a['out_dirs'], rtol=1e-6, atol=1e-7, what='dirs')
            assert_close(to, a['out_origins'], rtol=1e-6, atol=1e-7, what='origins')
        elif fn == 'fancy_integration':
            kw = {k: cfg[k] for k in ('clamp_mode', 'last_back', 'white_back', 'max_depth', 'fill_mode') if k in cfg}
            noise = t(a['in_noise']) if 'in_noise' in a else None
            rgb, depth, w = vr.fancy_integration(t(a['in_rs']).clone(), t(a['in_d']), t(a['in_z']), 'cpu', noise_std=cfg.get('noise_std', 0),
                                                 noise=noise, **kw)
            assert_close(rgb, a['out_rgb'], rtol=1e-5, atol=1e-5, what=f'rgb {cfg}')
            assert_close(depth, a['out_depth'], rtol=1e-5, atol=1e-5, what='depth')
            assert_close(w, a['out_w'], rtol=1e-5, atol=1e-6, what='weights')
        elif fn == 'sample_pdf':
            s = vr.sample_pdf(t(a['in_bins']), t(a['in_w']), cfg['N_importance'], det=True)
            assert_close(s, a['out_samples'], rtol=1e-6, atol=1e-6, what='sample_pdf')
        elif fn == 'sample_pdf_rand':
            s = vr.sample_pdf(t(a['in_bins']), t(a['in_w']), cfg['N_importance'], u=t(a['in_u']))
            assert_close(s, a['out_samples'], rtol=1e-6, atol=1e-6, what='sample_pdf rand')
            torch.manual_seed(77)         # the call draws `torch.rand(rays, N)` first, like the reference
            s = vr.sample_pdf(t(a['in_bins']), t(a['in_w']), cfg['N_importance'], det=False)
            assert_close(s, a['out_samples'], rtol=1e-6, atol=1e-6, what='sample_pdf det=False')
    with pytest.raises(ValueError):
        vr.fancy_integration(torch.zeros(1, 1, 2, 4), torch.ones(1, 1, 3), torch.zeros(1, 1, 2, 1), 'cpu', noise_std=0, clamp_mode=None)


def test_triplane_cpu(golden):
    from dnnlib import util
    for cfg, a in golden('triplane'):
        assert_close(util.sample_from_triplane(t(a['in_coords']), t(a['in_grid'])), a['out_feat'], rtol=1e-6, atol=1e-6, what=str(cfg))


def test_networks_cpu(golden):
    from training import networks
    from torch_utils.ops import upfirdn2d
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    for cfg, a in golden('networks').select(fn='modulated_conv2d'):
        kw = {k: cfg[k] for k in ('up', 'padding', 'demodulate', 'fused_modconv', 'flip_weight')}
        y = networks.modulated_conv2d(x=t(a['in_x']), weight=t(a['in_w']), styles=t(a['in_s']), noise=t(a['in_noise']),
                                      resample_filter=(f if cfg['up'] > 1 else None), **kw)
        assert_close(y, a['out_y'], rtol=1e-4, atol=1e-5, what=str(cfg))
    (cfg, a), = golden('networks').select(fn='mapping')
    m = networks.MappingNetwork(z_dim=16, c_dim=25, w_dim=12, num_ws=5, num_layers=3).eval()
    m.load_state_dict({k[len('sd_mapping.'):]: t(v) for k, v in a.items() if k.startswith('sd_mapping.')})
    with torch.no_grad():
        assert_close(m(t(a['in_z']), t(a['in_c'])), a['out_ws'], rtol=1e-5, atol=1e-6, what='mapping')
        assert_close(m(t(a['in_z']), t(a['in_c']), truncation_psi=0.6, truncation_cutoff=3), a['out_ws_trunc'], rtol=1e-5, atol=1e-6, what='trunc')


def load_golden_generator(golden, device='cpu'):
    from training import triplane
    (cfg, a), = golden('generator_tiny').cases
    G = triplane.TriPlaneGenerator(triplane.tiny_spec()).eval()
    sd = {k[len('sd_'):]: t(v) for k, v in a.items() if k.startswith('sd_')}
    missing, unexpected = G.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return G.to(device), cfg, a


def test_generator_matches_reference_assembly_cpu(golden):
    """Product generator (CPU path) == the same topology assembled from the reference's own modules."""
    G, cfg, a = load_golden_generator(golden)
    assert G.num_ws == a['out_ws'].shape[1] and G.synthesis.num_ws == G.num_ws
    with torch.no_grad():
        ws = G.mapping(t(a['in_z']), t(a['in_c_cond']), truncation_psi=cfg['truncation_psi'])
        assert_close(ws, a['out_ws'], rtol=1e-5, atol=1e-6, what='ws')
        out = G.synthesis(t(a['out_ws']), c=t(a['in_c']), noise_mode='const', ray_jitter=t(a['in_jitter']), return_dict=True)
    assert_close(out['planes'][0], a['out_img_v'], rtol=1e-4, atol=1e-4, what='texture tri-plane')
    assert_close(out['planes'][1], a['out_seg_v'], rtol=1e-4, atol=1e-4, what='semantic tri-plane')
    assert_close(out['image_depth'], a['out_depth'], rtol=1e-5, atol=1e-5, what='depth')
    assert_close(out['image'], a['out_img'], rtol=1e-3, atol=1e-3, what='image')
    assert_close(out['image_seg'], a['out_seg'], rtol=1e-3, atol=1e-3, what='seg')
    # driver-facing attributes
    syn = G.synthesis
    assert syn.voxel_block_resolutions == [4, 8, 16, 32] and syn.block_resolutions == [32, 64] and syn.render_size == 8
    assert syn.vb4.num_conv == 1 and syn.vb8.num_conv == 2 and syn.vb8.num_torgb == 1
    with torch.no_grad():
        sv = syn.renderer.sample_voxel(t(a['out_img_v']), t(a['out_seg_v']), t(a['in_sample_pts']))
    assert_close(sv, a['out_sample_out'], rtol=1e-4, atol=1e-5, what='sample_voxel')
    img, seg = G.synthesis(t(a['out_ws']), c=t(a['in_c']), ray_jitter=t(a['in_jitter']), return_seg=True)
    assert img.shape == (2, 3, 64, 64) and seg.shape == (2, 5, 64, 64)


def test_hierarchical_pass_cpu(golden):
    """Importance pass of the renderer (CPU step-wise path) == the oracle's composition of the same reference functions."""
    from oracle import generator as ogen, spec as ospec
    G, cfg, a = load_golden_generator(golden)
    sp = G.synthesis.renderer.spec
    n, rays, steps = 2, sp.render_size ** 2, sp.num_steps
    u = torch.rand(n * rays, steps, generator=torch.Generator().manual_seed(5))
    cam = t(a['in_c'])[:, :16].reshape(-1, 4, 4)
    sd = {k: v.detach() for k, v in G.state_dict().items()}
    want = ogen.render(sd, ospec.tiny(), t(a['out_img_v']), t(a['out_seg_v']), cam, jitter=t(a['in_jitter']), hierarchical=True,
                       importance_u=u)
    with torch.no_grad():
        got = G.synthesis.renderer(t(a['out_img_v']), t(a['out_seg_v']), cam, jitter=t(a['in_jitter']), hierarchical=True, importance_u=u)
        base = G.synthesis.renderer(t(a['out_img_v']), t(a['out_seg_v']), cam, jitter=t(a['in_jitter']))
    for g_, w_, name in zip(got, want, ('features', 'depth', 'weight sum')):
        assert_close(g_, w_, rtol=1e-4, atol=1e-5, what=f'hierarchical {name}')
    assert float((got[0] - base[0]).abs().max()) > 1e-4, 'the importance pass must change the result'
    out = G.synthesis(t(a['out_ws']), c=t(a['in_c']), ray_jitter=t(a['in_jitter']), render_params=dict(hierarchical=True, importance_u=u),
                      return_dict=True)
    assert_close(out['image_depth'], want[1], rtol=1e-4, atol=1e-5, what='hierarchical through G.synthesis')


def test_style_plan_matches_block_forward(golden):
    """`networks._style_plan` (what the side-stream style prefetch enumerates) lists exactly the (layer, w) pairs a block's
    forward consumes, in order — checked by recording the real calls of a CPU forward pass."""
    from training import networks
    G, cfg, a = load_golden_generator(golden)
    syn = G.synthesis
    ws = t(a['out_ws'])
    voxel_ws, block_ws = syn.split_ws(ws)
    planned = []
    for res, w in list(zip(syn.voxel_block_resolutions, voxel_ws)):
        planned += [(kind, id(mod.affine) if kind == 'conv' else id(mod.torgb.affine), wl.data_ptr())
                    for kind, mod, wl in networks._style_plan(getattr(syn, f'vb{res}'), w)]
    for res, w in list(zip(syn.block_resolutions, block_ws)):
        planned += [(kind, id(mod.affine) if kind == 'conv' else id(mod.torgb.affine), wl.data_ptr())
                    for kind, mod, wl in networks._style_plan(getattr(syn, f'b{res}'), w)]
    seen = []
    real, real_heads = networks._styles_and_dcoefs, networks._dual_head
    conv_affines = {p_[1] for p_ in planned if p_[0] == 'conv'}

    def spy(affine, w, weight, demodulate):
        if id(affine) in conv_affines:
            seen.append(('conv', id(affine), w.data_ptr()))
        return real(affine, w, weight, demodulate)

    def spy_heads(x, torgb, toseg, w):
        seen.append(('heads', id(torgb.affine), w.data_ptr()))
        return real_heads(x, torgb, toseg, w)
    networks._styles_and_dcoefs, networks._dual_head = spy, spy_heads
    try:
        with torch.no_grad():
            syn(ws, c=t(a['in_c']), noise_mode='const', ray_jitter=t(a['in_jitter']))
    finally:
        networks._styles_and_dcoefs, networks._dual_head = real, real_heads
    assert planned == seen


def test_shape_extraction_cpu(golden):
    """extract_shapes.py lattice (bit-exact vs the reference-run fixture) and chunked density query == oracle (tiny G)."""
    from training import shape_extraction as se, triplane
    for cfg, a in golden('post').select(fn='create_samples'):
        s, origin, size = se.create_samples(cfg['N'], voxel_origin=cfg.get('voxel_origin', (0, 0, 0)), cube_length=cfg['cube_length'])
        assert np.array_equal(s.numpy(), a['out_samples'])
        if 'out_origin' in a:
            assert np.array_equal(origin, a['out_origin']) and size == cfg['voxel_size']
    (cfg, a), = golden('generator_tiny').cases
    G = triplane.TriPlaneGenerator(triplane.tiny_spec()).eval()
    sd = {k[len('sd_'):]: t(v) for k, v in a.items() if k.startswith('sd_')}
    G.load_state_dict(sd)
    z, c = t(a['in_z'])[:1], t(a['in_c_cond'])[:1]
    N = 10
    cube = se.sample_generator_ide3d(G, None, z, c, max_batch=300, voxel_resolution=N, cube_length=1.0, psi=cfg['truncation_psi'], noise_mode='const')
    assert cube.shape == (N, N, N) and cube.dtype == np.float32
    ws = ogen.mapping(sd, ospec.tiny(), z, c, truncation_psi=cfg['truncation_psi'], ops=fast_ops)
    planes = ogen.backbone(sd, ospec.tiny(), ws, 'const', fast_ops)
    ref = ogen.sample_voxel(sd, ospec.tiny(), planes[0], planes[1], 0.9 * oracle_ops.create_samples(N, cube_length=1.0), fast_ops)[:, -1]
    assert_close(torch.from_numpy(cube).reshape(-1), ref, rtol=1e-4, atol=1e-5, what='density cube')
    one = se.sample_generator_ide3d(G, None, z, c, max_batch=None, voxel_resolution=N, cube_length=1.0, psi=cfg['truncation_psi'], noise_mode='const')
    assert np.array_equal(one, cube)


def test_density_cube_files(tmp_path):
    """`.npy` + MRC2014 (mode 2) sinks of extract_shapes.py:191-194: header fields and voxel round trip."""
    import struct
    from training import shape_extraction as se
    v = np.random.RandomState(3).randn(5, 6, 7).astype(np.float32)
    se.save_density_cube(str(tmp_path), '17', v)
    assert np.array_equal(np.load(tmp_path / '17.npy'), v)
    raw = (tmp_path / '17.mrc').read_bytes()
    assert len(raw) == 1024 + v.nbytes
    nx, ny, nz, mode = struct.unpack_from('<4i', raw, 0)
    assert (nx, ny, nz, mode) == (7, 6, 5, 2) and struct.unpack_from('<3i', raw, 64) == (1, 2, 3)
    assert raw[208:212] == b'MAP ' and struct.unpack_from('<i', raw, 108)[0] == 20140
    dmin, dmax, dmean = struct.unpack_from('<3f', raw, 76)
    assert np.isclose(dmin, v.min()) and np.isclose(dmax, v.max()) and np.isclose(dmean, v.mean(), atol=1e-6)
    assert np.array_equal(se.read_mrc(tmp_path / '17.mrc'), v)


def test_video_sweep_cpu(golden):
    """gen_videos.py frame loop (training.video_render): batched cells + cached tri-planes == the reference's schedule
    rendered cell by cell (batch 1, no caching), camera sweep == direct LookAtPoseSampler calls, ws interpolation == scipy."""
    import scipy.interpolate
    from training import video_render as vr_, triplane, distributed_render as dr
    from training.volumetric_rendering import LookAtPoseSampler
    (cfg, a), = golden('generator_tiny').cases
    G = triplane.TriPlaneGenerator(triplane.tiny_spec()).eval()
    G.load_state_dict({k[len('sd_'):]: t(v) for k, v in a.items() if k.startswith('sd_')})
    seeds, w_frames, grid = [3, 5, 7, 11], 3, (2, 2)
    frames = list(vr_.gen_interp_frames(G, seeds, w_frames=w_frames, grid_dims=grid, psi=0.7, truncation_cutoff=None, device=torch.device('cpu'),
                                       ray_jitter=False))
    assert len(frames) == w_frames and frames[0].shape == (2 * 64, 2 * 128, 3) and frames[0].dtype == torch.uint8
    lookat = torch.tensor([0, 0, 0.2])
    front = LookAtPoseSampler.sample(math.pi / 2, math.pi / 2, lookat, radius=2.7)
    intr = torch.tensor(vr_.INTRINSICS, dtype=torch.float32).reshape(1, 9)
    c0 = torch.cat([front.reshape(1, 16), intr], 1)
    with torch.no_grad():
        for f in range(w_frames):
            ang = 2 * math.pi * f / w_frames
            pose = LookAtPoseSampler.sample(math.pi / 2 - 0.5 * np.sin(ang), math.pi / 2 - 0.05 + 0.25 * np.cos(ang), lookat, radius=2.7)
            c = torch.cat([pose.reshape(1, 16), intr], 1)
            assert torch.equal(vr_.sweep_pose(f, w_frames, lookat), c)
            cells = []
            for s in seeds:
                z = torch.from_numpy(np.random.RandomState(s).randn(1, G.z_dim)).float()
                ws = G.mapping(z, c0, truncation_psi=0.7)
                img, seg = G.synthesis(ws, c=c, noise_mode='const', return_seg=True, ray_jitter=False)
                cells.append(dr.frames_u8(img, seg)[0])
            ref = torch.cat([torch.cat(cells[:2], dim=1), torch.cat(cells[2:], dim=1)], dim=0)
            diff = (frames[f].int() - ref.int()).abs()
            # RGB differs by at most 1 LSB from batch-composition rounding; a flipped arg-max at a class boundary recolours a pixel
            assert (diff > 1).float().mean() < 5e-3, f'frame {f}: {(diff > 1).float().mean():.4f} of the bytes differ by more than 1'
    # interpolation of two keyframes == scipy on the tiled sequence
    g = np.random.RandomState(0).randn(2, 4, 6).astype(np.float32)
    got = vr_.interpolate_ws(g, 4)
    x = np.arange(-4, 6); ref_i = scipy.interpolate.interp1d(x, np.tile(g, [5, 1, 1]), kind='cubic', axis=0)
    assert got.shape == (8, 4, 6) and np.allclose(got[5], ref_i(5 / 4), atol=1e-6) and np.allclose(got[0], g[0], atol=1e-6)


def _fingerprint(module):
    out = []
    for k, v in module.state_dict().items():
        v = v.double().flatten()
        out.append([float(v.abs().sum()), float((v * torch.linspace(0.5, 1.5, v.numel(), dtype=torch.float64)).sum())])
    return np.array(out)


def build_encoder(cfg):
    """Encoder of a golden case, built under the case's seed (same parameter creation order as the reference)."""
    from training import encoders
    torch.manual_seed(cfg['seed'])
    if cfg['fn'] == 'hybrid_encoder':
        return encoders.HybridEncoder(size=cfg['size'], n_latents_app=cfg['n_latents_app'], n_latents_geo=cfg['n_latents_geo'],
                                      w_dim=cfg['w_dim'], add_dim=cfg['add_dim']).eval()
    return encoders.Encoder(size=cfg['size'], n_latents=cfg['n_latents'], w_dim=cfg['w_dim'], add_dim=cfg['add_dim']).eval()


def test_y4m_video_sink(tmp_path):
    """Dependency-free video sink (training.video_render.write_y4m): header, frame count, BT.601 limited-range conversion
    checked on the primaries and against the float definition."""
    from training import video_render
    g = torch.Generator().manual_seed(2)
    frames = [torch.randint(0, 256, (6, 10, 3), generator=g, dtype=torch.uint8) for _ in range(3)]
    frames[0][0, 0] = torch.tensor([255, 255, 255], dtype=torch.uint8); frames[0][0, 1] = torch.tensor([0, 0, 0], dtype=torch.uint8)
    frames[0][0, 2] = torch.tensor([255, 0, 0], dtype=torch.uint8)
    path = str(tmp_path / 'clip.y4m')
    assert video_render.write_y4m(iter(frames), path, fps=30) == 3
    info, yuv = video_render.read_y4m(path)
    assert info['W'] == '10' and info['H'] == '6' and info['F'] == '30:1' and yuv.shape == (3, 3, 6, 10)
    assert tuple(yuv[0, :, 0, 0]) == (235, 128, 128) and tuple(yuv[0, :, 0, 1]) == (16, 128, 128)      # white, black
    assert tuple(yuv[0, :, 0, 2]) == (82, 90, 240)                                                     # red
    rgb = torch.stack(frames).double()
    m = torch.tensor([[65.481, 128.553, 24.966], [-37.797, -74.203, 112.0], [112.0, -93.786, -18.214]], dtype=torch.float64) / 255
    want = (rgb @ m.t()) + torch.tensor([16.0, 128.0, 128.0], dtype=torch.float64)
    assert float((torch.from_numpy(yuv.astype(np.float64)).permute(0, 2, 3, 1) - want).abs().max()) <= 1.0


def test_encoders_cpu(golden):
    """HybridEncoder / Encoder (inversion/networks.py:1559-1665): same state-dict keys and (seeded) weights as the
    reference, same outputs as the reference run that produced the fixture."""
    for cfg, a in golden('encoder'):
        E = build_encoder(cfg)
        assert list(E.state_dict().keys()) == cfg['keys']
        np.testing.assert_allclose(_fingerprint(E), a['sd_fingerprint'], rtol=1e-12, atol=0)
        with torch.no_grad():
            if cfg['fn'] == 'hybrid_encoder':
                assert_close(E(t(a['in_img']), t(a['in_seg'])), a['out_ws'], rtol=1e-4, atol=1e-4, what='hybrid encoder ws')
            else:
                ws, extra = E(t(a['in_x']))
                assert_close(ws, a['out_ws'], rtol=1e-4, atol=1e-4, what='encoder ws')
                assert_close(extra, a['out_extra'], rtol=1e-4, atol=1e-4, what='encoder extra head')


def test_full_spec_shapes():
    from training import triplane
    sp = triplane.GeneratorSpec()
    assert sp.voxel_resolutions() == [4, 8, 16, 32, 64, 128, 256]
    assert [sp.voxel_width(r) for r in sp.voxel_resolutions()] == [512, 512, 512, 512, 512, 256, 128]
    assert sp.sr_resolutions() == [256, 512] and sp.sr_widths() == {256: 128, 512: 64}
    c = triplane.conditioning_label()
    assert c.shape == (1, 25) and float(c[0, 11]) == pytest.approx(2.7)
    assert_close(triplane.camera_label(0.0)[:, :16], c[:, :16], rtol=0, atol=1e-6, what='frontal pose == conditioning pose')


def test_dnnlib_util_lazy_reference_import_failure_is_attribute_error(tmp_path):
    """a reference dnnlib/util.py that fails to import (it imports `requests` ... at module level) must surface as
    AttributeError from the overlay's module `__getattr__` — every time, not only on the first access — with the cause chained."""
    code = r'''
import os, sys
overlay, fake = sys.argv[1], sys.argv[2]
os.makedirs(os.path.join(fake, 'dnnlib'))
open(os.path.join(fake, 'dnnlib', '__init__.py'), 'w').write('')
open(os.path.join(fake, 'dnnlib', 'util.py'), 'w').write('import a_module_that_does_not_exist_anywhere\n')
sys.path[:0] = [overlay, fake]
import dnnlib, dnnlib.util as u
assert any(os.path.abspath(d) == os.path.join(fake, 'dnnlib') for d in dnnlib.__path__)
for attempt in range(3):                      # the failure is cached: same answer on every access
    assert not hasattr(u, 'format_time')
    assert getattr(u, 'Logger', 'dflt') == 'dflt'
    try:
        u.make_cache_dir_path
    except AttributeError as e:
        assert isinstance(e.__cause__, ImportError) and isinstance(e.__cause__.__cause__, ModuleNotFoundError), repr(e.__cause__)
    else:
        raise SystemExit('expected AttributeError')
assert callable(u.sample_from_triplane) and u.EasyDict(a=1).a == 1          # what the overlay states itself keeps working
print('LAZY_OK')
'''
    import subprocess
    import sys
    res = subprocess.run([sys.executable, '-c', code, os.path.join(ROOT, 'ide-3d_amd'), str(tmp_path / 'fake_ref')],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and 'LAZY_OK' in res.stdout, res.stdout[-3000:]


def test_row_floats_readable_is_the_distance_to_the_end_of_storage():
    """`ide3d_upfirdn2d_params.x_row_floats` (ABI 5): what the binding promises the kernel may read from the start of the row that starts
    last in storage — the 16-byte staging path is only taken when round_up(in_w, 4) fits into it."""
    from torch_utils import hip_plugin
    base = torch.zeros(2, 3, 9, 20)
    assert hip_plugin._row_floats_readable(base) == 20                       # dense: exactly one row
    assert hip_plugin._row_floats_readable(base[..., 4:]) == 16              # W-offset view: the last row ends the storage
    assert hip_plugin._row_floats_readable(base[..., :13]) == 20             # narrow view: the rest of the row is readable
    assert hip_plugin._row_floats_readable(base[:, :, ::2]) == 20            # H-strided, odd row count: last viewed row = last row
    assert hip_plugin._row_floats_readable(base[:, :, :8:2]) == 20 + 2 * 20  # H-strided view ending two rows earlier
    assert hip_plugin._row_floats_readable(base.flip(3)) == 0 if any(s <= 0 for s in base.flip(3).stride()) else True
    assert hip_plugin._row_floats_readable(torch.zeros(4, 5)) == 0           # not rank 4: no promise


def test_face_parser_state_dict_layout_and_cpu_path(golden):
    """training/face_parsing.py: the state dict has the reference BiSeNet's keys and shapes (so `segNet-20Class.pth`, dnnlib/seg_tools.py:128,
    loads), and its CPU forward reproduces the reference run of tests/golden/bisenet.npz; the label pipeline (argmax, id_remap, scatter) too."""
    from training import face_parsing
    from oracle import face_parsing as ofp
    cases = golden('bisenet').select(fn='bisenet')
    shapes = cases[0][0]['shapes']
    net = face_parsing.BiSeNet(n_classes=20).eval()
    mine = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert list(mine) == list(shapes), 'state dict keys (and their order) differ from the reference BiSeNet'
    assert mine == shapes
    net.load_state_dict(ofp.synthetic_state_dict(shapes))
    for cfg, a in cases:
        with torch.no_grad():
            out, aux16, aux32 = net(t(a['in_x']))
        assert aux16 is None and aux32 is None
        assert_close(out, a['out_logits'], rtol=0, atol=2e-5 * float(np.abs(a['out_logits']).max()), what='BiSeNet logits (CPU path)')
    (cfg, a), = golden('bisenet').select(fn='labels')
    logits = t(cases[1][1]['out_logits'])
    seg = face_parsing.id_remap(logits.argmax(1, keepdim=True), 'celebahq')
    assert np.array_equal(seg.numpy().astype(np.uint8), a['out_remap'])
    assert np.array_equal(face_parsing.scatter(seg, label_size=tuple(logits.shape[2:])).numpy().astype(np.uint8), a['out_onehot'])
    # parsing_img / face_parsing shapes (dnnlib/seg_tools.py:100-123): any input size -> one-hot at 512 x 512
    img, onehot = face_parsing.parsing_img(lambda x: (logits,), t(cases[1][1]['in_x']), return_mask=False)
    assert onehot.shape == (1, 1, 96, 64)
