"""Parity at the configurations that are BENCHMARKED (BASELINE.json configs 2, 3 and 5), not at reduced sizes:

  config 2  full-size ide3d-ffhq-64-512, batch 4, hipGraph replay (`GraphedRenderer`, side-stream style prefetch, image-batched
            tiles) — exactly what `bench.py` times: all four images vs the CPU oracle, and graph == eager bit for bit;
  config 3  one full-size 2x2 `image_seg` grid frame of `training.video_render` (cached tri-planes, uint8 colour mapping and
            layout on the device) vs frames built from the oracle;
  config 5  the 256^3 `extract_shapes.py` lattice: device lattice bit-equal to `create_samples(256)` (the float-division quirk
            is exact at 256 only because 256^3 = 2^24 — this is the case that has to be run), sigma of the single-launch cube
            vs the oracle on a strided subset;
  plus the fall-back of the renderer for decoder shapes the fused kernel is not compiled for.

Tolerances: config 2 (all fp32-grade arithmetics): tri-planes, final images and seg logits 2e-5 of the tensor's scale, raw 64x64 image 1e-4
(measured values: gpurun_out/parity_measured.json); configs 3-5 as stated in the tests; uint8 frames within 1 LSB except < 0.5 % of pixels, seg argmax flips < 0.5 %.
`pytest -m gpu`.
"""

import math
import os

import numpy as np
import pytest
import torch

from oracle import fast_ops
from oracle import generator as ogen
from oracle import ops as oracle_ops
from oracle import spec as ospec

pytestmark = pytest.mark.gpu

BENCH_YAWS = (-0.5, 0.0, 0.5, 0.25)          # bench.py's camera labels


def _calls(name):
    from torch_utils import hip_plugin
    return hip_plugin.CALLS.get(name, 0)


def _rel(actual, expected, tol, what):
    a = actual.detach().cpu().double(); e = torch.as_tensor(expected).detach().cpu().double()
    assert a.shape == e.shape, f'{what}: shape {tuple(a.shape)} != {tuple(e.shape)}'
    scale = float(e.abs().max()) + 1e-12
    err = float((a - e).abs().max())
    assert err <= tol * scale, f'{what}: max abs err {err:.3e} > {tol} * scale {scale:.3e}'


@pytest.fixture(scope='module')
def oracle_threads():
    """The fp32 torch-CPU oracle is fastest with a few dozen threads, not with one per core of a 256-thread host."""
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    yield
    torch.set_num_threads(old)


@pytest.fixture(scope='module')
def bench_generator(gpu_device):
    """The generator bench.py builds: torch.manual_seed(0) random init of the full-size spec (+ its CPU state dict)."""
    from training import triplane
    torch.manual_seed(0)
    G = triplane.TriPlaneGenerator().eval()
    sd = {k: v.detach().clone() for k, v in G.state_dict().items()}
    return G.to(gpu_device), sd


MEASURED = {}          # max relative errors observed in this session (written to gpurun_out/parity_measured.json by the last test of the file)


def _rel_m(actual, expected, tol, what, key=None):
    a = actual.detach().cpu().double(); e = torch.as_tensor(expected).detach().cpu().double()
    assert a.shape == e.shape, f'{what}: shape {tuple(a.shape)} != {tuple(e.shape)}'
    scale = float(e.abs().max()) + 1e-12
    err = float((a - e).abs().max())
    if key is not None:
        MEASURED[key] = max(MEASURED.get(key, 0.0), err / scale)
    assert err <= tol * scale, f'{what}: max abs err {err:.3e} > {tol} * scale {scale:.3e} (rel {err / scale:.2e})'


FRAME_TOL = 2e-5       # the float tolerance of the full-frame comparisons below, of the tensor's scale


def _frame_u8_check(got, img_ref, seg_ref, what, tol=FRAME_TOL):
    """uint8 frame `got` [H, 2W, 3] (RGB | palette[argmax seg]) against the ORACLE'S FLOATS img_ref [1, 3, H, W], seg_ref [1, K, H, W]:
      * every RGB byte within 1 LSB of the oracle's byte, and a 1-LSB difference ONLY where the oracle's own value x = img * 127.5 + 128
        lies within tol * scale * 127.5 of the integer it truncates at (two float images that agree to tol * scale can only straddle a
        truncation boundary that close to one of them);
      * a different class colour ONLY where the oracle's two largest logits are closer than 2 * tol * scale.
    With the measured float error (4e-6 of scale) anything else is a defect, not rounding."""
    img_ref = torch.as_tensor(img_ref).detach().cpu().float(); seg_ref = torch.as_tensor(seg_ref).detach().cpu().float()
    h, w = img_ref.shape[-2:]
    x = (img_ref[0] * 127.5 + 128).permute(1, 2, 0).numpy()
    want = np.clip(x, 0, 255).astype(np.uint8)
    d = got[:, :w].astype(np.int32) - want.astype(np.int32)
    assert np.abs(d).max() <= 1, f'{what}: RGB byte off by {np.abs(d).max()} LSB'
    margin = tol * float(img_ref.abs().max()) * 127.5
    unexplained = (d != 0) & (np.abs(x - np.round(x)) > margin)
    assert not unexplained.any(), (f'{what}: {int(unexplained.sum())} RGB bytes differ where the oracle is more than {margin:.2e} from a truncation boundary '
                                   f'({int((d != 0).sum())} 1-LSB sites in all)')
    top2 = torch.topk(seg_ref[0], 2, dim=0).values.numpy()
    gap = top2[0] - top2[1]
    want_col = oracle_ops.frame_u8(img_ref, seg_ref)[0][:, w:]
    flips = (got[:, w:] != want_col).any(axis=-1)
    unexplained = flips & (gap > 2 * tol * float(seg_ref.abs().max()))
    assert not unexplained.any(), f'{what}: {int(unexplained.sum())} class flips where the oracle\'s top-2 logit gap exceeds 2 * tol * scale ({int(flips.sum())} flips in all)'
    return int((d != 0).sum()), int(flips.sum())


def _config2_inputs():
    from training import triplane
    B = 4
    z = torch.from_numpy(np.stack([np.random.RandomState(s).randn(512) for s in range(B)]))
    cams = torch.cat([triplane.camera_label(y) for y in BENCH_YAWS])
    cond = triplane.conditioning_label().repeat(B, 1)
    jit = torch.rand(B, 4096, 96, generator=torch.Generator().manual_seed(11))
    return B, z, cams, cond, jit


@pytest.fixture(scope='module')
def config2_oracle(bench_generator, oracle_threads):
    """The CPU oracle's outputs for the four images bench.py's parity step renders (computed once for all arithmetics)."""
    _G, sd = bench_generator
    B, z, cams, cond, jit = _config2_inputs()
    osp = ospec.Spec()
    refs = []
    for i in range(B):          # one oracle image at a time (memory)
        ws_o = ogen.mapping(sd, osp, z[i:i + 1], cond[i:i + 1], ops=fast_ops)
        ref = ogen.synthesis(sd, osp, ws_o, cams[i:i + 1], jitter=jit[i:i + 1], ops=fast_ops)
        refs.append(dict(ws=ws_o, planes=[p.clone() for p in ref['planes']], image_raw=ref['image_raw'], image=ref['image'], image_seg=ref['image_seg'],
                         frame=oracle_ops.frame_u8(ref['image'], ref['image_seg'])[0]))
    return refs


@pytest.mark.parametrize('arith', ['fp32', 'bf16x6', 'f16x3'])
def test_config2_batch4_graph_replay_vs_oracle(bench_generator, config2_oracle, gpu_device, arith):
    """BASELINE config 2 exactly as bench.py times it (batch 4, captured hipGraph), FULL frames, with exact fp32 products, in the library-default
    bf16x6 and in f16x3: every image vs the CPU oracle at 2e-5 of the tensor's scale (measured 3e-6;
    the round-3 bound was 2e-3), graph replay == eager launches bit for bit, uint8 frames within 1 LSB."""
    from torch_utils import hip_plugin
    from training import triplane
    from training import distributed_render as dr
    G, _sd = bench_generator
    B, z, cams, cond, jit = _config2_inputs()
    try:
        process_arith = hip_plugin.conv_arithmetic()
        run = triplane.GraphedRenderer(G, B, gpu_device, conv_arithmetic=arith)
        assert hip_plugin.conv_arithmetic() == process_arith, 'GraphedRenderer(conv_arithmetic=) must restore the process setting'
        # a replay with other inputs first: the compared replay must not depend on what the capture / previous call left behind
        run(torch.randn(B, 512, device=gpu_device), cond.to(gpu_device), cams.flip(0).to(gpu_device))
        img_g, seg_g = run(z.to(gpu_device), cond.to(gpu_device), cams.to(gpu_device), jitter=jit.to(gpu_device))
        img_g, seg_g = img_g.clone(), seg_g.clone()
        assert img_g.shape == (B, 3, 512, 512) and seg_g.shape == (B, 19, 512, 512)

        hip_plugin.conv_arithmetic(arith)
        before = _calls('render_rays')
        with torch.no_grad():
            ws = G.mapping(z.to(gpu_device).float(), cond.to(gpu_device))
            out = G.synthesis(ws, c=cams.to(gpu_device), noise_mode='const', ray_jitter=jit.to(gpu_device), return_dict=True)
        assert _calls('render_rays') == before + 1, 'the fused HIP ray-marcher must have run'
    finally:
        hip_plugin.conv_arithmetic('default')
    assert torch.equal(img_g, out['image']) and torch.equal(seg_g, out['image_seg']), f'{arith}: hipGraph replay != eager launches at full size'

    frames_gpu = dr.frames_u8(img_g, seg_g, dr.palette_tensor(19, gpu_device)).cpu().numpy()
    for i, ref in enumerate(config2_oracle):
        _rel_m(ws[i:i + 1], ref['ws'], 2e-5, f'ws[{i}]', f'{arith}/ws')
        _rel_m(out['planes'][0][i:i + 1], ref['planes'][0], 2e-5, f'{arith}: texture tri-plane [{i}]', f'{arith}/planes_tex')
        _rel_m(out['planes'][1][i:i + 1], ref['planes'][1], 2e-5, f'{arith}: semantic tri-plane [{i}]', f'{arith}/planes_seg')
        _rel_m(out['image_raw'][i:i + 1], ref['image_raw'], 1e-4, f'{arith}: raw 64x64 image [{i}]', f'{arith}/image_raw')
        _rel_m(img_g[i:i + 1], ref['image'], 2e-5, f'{arith}: image 512 [{i}]', f'{arith}/image')
        _rel_m(seg_g[i:i + 1], ref['image_seg'], 2e-5, f'{arith}: seg 512 [{i}]', f'{arith}/image_seg')
        lsb, flips = _frame_u8_check(frames_gpu[i], ref['image'], ref['image_seg'], f'{arith}: uint8 frame [{i}]')
        MEASURED[f'{arith}/frame_1lsb_sites'] = MEASURED.get(f'{arith}/frame_1lsb_sites', 0) + lsb
        MEASURED[f'{arith}/frame_class_flips'] = MEASURED.get(f'{arith}/frame_class_flips', 0) + flips


def test_dropin_batch1_eager_equals_graphed_batch4_row(bench_generator, gpu_device):
    """What a drop-in caller of gen_images.py:88-114 runs — batch 1, eager launches, library-default arithmetic — against the row of the
    benchmarked batch-4 graph for the same seed / camera / jitter.  Kernel selection depends on the batch (split-K planning follows the
    number of workgroups), so the two may differ in summation order: bit equality is reported (MEASURED), closeness at fp32 rounding
    level (1e-5 of the scale) is required."""
    from training import triplane
    G, _sd = bench_generator
    B, z, cams, cond, jit = _config2_inputs()
    dev = gpu_device
    run = triplane.GraphedRenderer(G, B, dev)                     # library-default arithmetic, like the eager calls below
    img4, seg4 = run(z.to(dev), cond.to(dev), cams.to(dev), jitter=jit.to(dev))
    img4, seg4 = img4.clone(), seg4.clone()
    equal = True
    for i in range(B):
        with torch.no_grad():
            ws = G.mapping(z[i:i + 1].to(dev).float(), cond[i:i + 1].to(dev))
            img1, seg1 = G.synthesis(ws, c=cams[i:i + 1].to(dev), noise_mode='const', return_seg=True, ray_jitter=jit[i:i + 1].to(dev))
        equal = equal and torch.equal(img1, img4[i:i + 1]) and torch.equal(seg1, seg4[i:i + 1])
        _rel_m(img1, img4[i:i + 1], 1e-5, f'batch-1 eager image [{i}] vs graphed batch-4 row', 'dropin_b1/image')
        _rel_m(seg1, seg4[i:i + 1], 1e-5, f'batch-1 eager seg [{i}] vs graphed batch-4 row', 'dropin_b1/image_seg')
    MEASURED['dropin_b1/bit_equal'] = bool(equal)


def test_f16x3_frame_with_weight_rows_and_columns_over_many_octaves(bench_generator, gpu_device, oracle_threads):
    """Frame-level stress of f16x3's block scaling: the full-size generator with every 3x3 convolution's weight ROWS
    scaled by 2^U(-12, 12) (the demodulation cancels a row scale exactly, so the function is unchanged but the per-row power-of-two scale
    chosen at pack time and the demodulation coefficients span 24 octaves) and its input COLUMNS by 2^U(-4, 4) (not cancelled: the
    operands x * s inside one image then span 8+ octaves, what a trained pickle's styles do) — one image vs the CPU oracle on the same
    weights, in f16x3 and in the fp32 default, same 2e-5 bound."""
    import copy
    from torch_utils import hip_plugin
    from training import triplane
    G0, _ = bench_generator
    G = copy.deepcopy(G0)
    gen = torch.Generator().manual_seed(5)
    n_scaled = 0
    with torch.no_grad():
        for name, prm in G.named_parameters():
            if name.endswith('.weight') and prm.ndim == 4 and prm.shape[-1] == 3:
                rows = torch.exp2(torch.randint(-12, 13, [prm.shape[0], 1, 1, 1], generator=gen).float()).to(prm.device)
                cols = torch.exp2(torch.randint(-4, 5, [1, prm.shape[1], 1, 1], generator=gen).float()).to(prm.device)
                prm.mul_(rows).mul_(cols)
                n_scaled += 1
    assert n_scaled >= 12, n_scaled
    sd = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
    _B, z, cams, cond, jit = _config2_inputs()
    osp = ospec.Spec()
    ws_o = ogen.mapping(sd, osp, z[:1], cond[:1], ops=fast_ops)
    ref = ogen.synthesis(sd, osp, ws_o, cams[:1], jitter=jit[:1], ops=fast_ops)
    dev = gpu_device
    try:
        for arith in ('fp32', 'f16x3'):
            hip_plugin.conv_arithmetic(arith)
            with torch.no_grad():
                ws = G.mapping(z[:1].to(dev).float(), cond[:1].to(dev))
                out = G.synthesis(ws, c=cams[:1].to(dev), noise_mode='const', ray_jitter=jit[:1].to(dev), return_dict=True)
            _rel_m(out['planes'][0], ref['planes'][0], 2e-5, f'{arith}: texture tri-plane, scaled weights', f'scaled_weights/{arith}/planes_tex')
            _rel_m(out['image'], ref['image'], 2e-5, f'{arith}: image, scaled weights', f'scaled_weights/{arith}/image')
            _rel_m(out['image_seg'], ref['image_seg'], 2e-5, f'{arith}: seg, scaled weights', f'scaled_weights/{arith}/image_seg')
    finally:
        hip_plugin.conv_arithmetic('default')


def test_config3_full_size_grid_frame_vs_oracle(bench_generator, gpu_device, oracle_threads):
    """gen_videos.py 2x2 grid, `image_seg`: frames 0 and 30 of a 120-frame sweep from `video_render` vs oracle-built frames."""
    from training import video_render
    G, sd = bench_generator
    seeds, total, psi, cutoff = [0, 1, 2, 3], 120, 0.7, 14
    want_idx = (0, 30)
    before = _calls('frame_u8')
    frames = {}
    for i, f in enumerate(video_render.gen_interp_frames(G, seeds, w_frames=total, grid_dims=(2, 2), psi=psi, truncation_cutoff=cutoff,
                                                         device=gpu_device, ray_jitter=False)):
        if i in want_idx:
            frames[i] = f.cpu().numpy()
        if i >= max(want_idx):
            break
    assert _calls('frame_u8') > before
    osp = ospec.Spec()
    lookat = torch.tensor([0, 0, 0.2])
    # frontal conditioning label of gen_videos.py:83-88 (host-side camera math, golden-pinned in tests/test_host_cpu.py)
    front = video_render.LookAtPoseSampler.sample(math.pi / 2, math.pi / 2, lookat, radius=2.7)
    c_front = torch.cat([front.reshape(1, 16), torch.tensor(video_render.INTRINSICS, dtype=torch.float32).reshape(1, 9)], 1)
    zs = torch.from_numpy(np.stack([np.random.RandomState(s).randn(512) for s in seeds]))
    for idx in want_idx:
        c = video_render.sweep_pose(idx, total, lookat)
        cells = []
        for k in range(4):
            ws_o = ogen.mapping(sd, osp, zs[k:k + 1], c_front, truncation_psi=psi, truncation_cutoff=cutoff, ops=fast_ops)
            ref = ogen.synthesis(sd, osp, ws_o, c, jitter=None, ops=fast_ops)
            cells.append((ref['image'], ref['image_seg']))
        got = frames[idx]
        assert got.shape == (1024, 2048, 3)
        for k, (img_ref, seg_ref) in enumerate(cells):                 # cell k of the 2x2 grid: row k // 2, column k % 2 (layout_grid, gen_videos.py:24-38)
            cell = got[(k // 2) * 512:(k // 2 + 1) * 512, (k % 2) * 1024:(k % 2 + 1) * 1024]
            _frame_u8_check(cell, img_ref, seg_ref, f'frame {idx}, cell {k}')


def test_two_live_video_jobs_on_one_generator_keep_their_own_planes(bench_generator, gpu_device):
    """`gen_interp_frames` keeps a job's static tri-planes in buffers that outlive the job (so the next job replays the captured pass);
    two LIVE jobs of one generator — interleaved iterators — must not share a pair: each frame equals the frame of the same job run alone."""
    from training import video_render
    G, _ = bench_generator
    kw = dict(w_frames=8, grid_dims=(1, 1), psi=0.7, truncation_cutoff=14, device=gpu_device, ray_jitter=False)
    alone = {s: [f.clone() for f in video_render.gen_interp_frames(G, [s], **kw)][:3] for s in (11, 12)}
    a, b = video_render.gen_interp_frames(G, [11], **kw), video_render.gen_interp_frames(G, [12], **kw)
    for i in range(3):
        fa, fb = next(a), next(b)
        assert torch.equal(fa, alone[11][i]), f'job A frame {i} was rendered from the other job\'s planes'
        assert torch.equal(fb, alone[12][i]), f'job B frame {i}'
    pool = video_render._plane_pool[G.synthesis]
    assert sum(1 for e in pool if e[1]) == 2
    a.close(); b.close()
    assert sum(1 for e in pool if e[1]) == 0          # both pairs handed back: the next job reuses one (and the pass captured on it)


def test_config4_sharded_items_full_size_vs_oracle(bench_generator, gpu_device, oracle_threads):
    """BASELINE config 4's item path where it runs (round 3): `render_grid_sharded(world=1)` at full size — 2 seeds x 8 camera poses,
    all poses of a seed from ONE cached backbone pass, batches of 4 items, fixed per-item jitter — against oracle frames of the same
    (seed, pose) pairs, in the default (fp32) arithmetic and in the one bench.py runs (f16x3)."""
    from torch_utils import hip_plugin
    from training import distributed_render as dr
    from training import triplane
    G, sd = bench_generator
    seeds = [3, 11]
    yaws = [float(y) for y in np.linspace(-0.5, 0.5, 8)]
    osp = ospec.Spec()
    cond = triplane.conditioning_label()
    want = []
    for si, seed in enumerate(seeds):
        z = torch.from_numpy(np.random.RandomState(seed).randn(1, 512))
        ws_o = ogen.mapping(sd, osp, z, cond, ops=fast_ops)
        for pi, yaw in enumerate(yaws):
            jit = torch.rand([64 * 64, 96], generator=torch.Generator().manual_seed(77 + dr.item_index(si, pi, len(yaws))))[None]
            ref = ogen.synthesis(sd, osp, ws_o, triplane.camera_label(yaw), jitter=jit, ops=fast_ops)
            want.append((ref['image'], ref['image_seg']))
    from training import graph_cache
    passes = lambda: graph_cache.STATS['eager'] + graph_cache.STATS['replay'] + sum(v for k, v in graph_cache.STATS.items() if k.startswith('ineligible'))
    try:
        for arith in ('fp32', 'f16x3'):
            hip_plugin.conv_arithmetic(arith)
            before = _calls('render_rays'), _calls('frame_u8'), passes()
            got = dr.render_grid_sharded(G, seeds, yaws, gpu_device, rank=0, world=1, batch=4, jitter_seed=77).cpu().numpy()
            # 16 items in batches of 4: four `G.synthesis` passes (eager, then captured and replayed: training/graph_cache.py), four frame launches
            assert passes() - before[2] == 4 + 2 and _calls('frame_u8') - before[1] == 4 and _calls('render_rays') > before[0]       # + the two seeds' `planes` calls
            assert got.shape == (16, 512, 1024, 3)
            for i in range(16):
                _frame_u8_check(got[i], want[i][0], want[i][1], f'{arith}, item {i}')
    finally:
        hip_plugin.conv_arithmetic('default')


def test_config5_lattice_256_and_density_vs_oracle(bench_generator, gpu_device, oracle_threads):
    """extract_shapes.py at voxel_resolution 256: the lattice, bit for bit, and the single-launch sigma cube."""
    from training import shape_extraction as se
    from training import triplane
    G, sd = bench_generator
    N = 256
    dev, origin, vsize = se.create_samples(N, (0, 0, 0), 2.0, device=gpu_device)
    want = oracle_ops.create_samples(N, (0, 0, 0), 2.0)
    assert dev.shape == want.shape == (1, N ** 3, 3)
    assert torch.equal(dev.cpu(), want), 'device lattice != extract_shapes.create_samples(256)'
    del dev
    z = torch.from_numpy(np.random.RandomState(0).randn(1, 512))
    cond = triplane.conditioning_label()
    osp = ospec.Spec()
    ws_o = ogen.mapping(sd, osp, z, cond, truncation_psi=0.5, ops=fast_ops)
    planes_o = ogen.backbone(sd, osp, ws_o, 'const', fast_ops)
    before = _calls('density_lattice')
    cube = se.sample_generator_ide3d(G, None, z.to(gpu_device), cond.to(gpu_device), max_batch=None, voxel_resolution=N,
                                     cube_length=2.0, psi=0.5, to_numpy=False, noise_mode='const')
    assert _calls('density_lattice') == before + 1, 'one launch for the whole cube'
    assert cube.shape == (N, N, N)
    chunked = se.sample_generator_ide3d(G, None, z.to(gpu_device), cond.to(gpu_device), max_batch=1000000, voxel_resolution=N,
                                        cube_length=2.0, psi=0.5, to_numpy=False, noise_mode='const')
    assert torch.equal(cube, chunked), 'chunked query loop != single launch'
    sub = torch.arange(0, N ** 3, 4099)
    pts = 0.9 * want[:, sub]
    ref = ogen.sample_voxel(sd, osp, planes_o[0], planes_o[1], pts, fast_ops)[:, -1]
    _rel(cube.reshape(-1)[sub.to(gpu_device)], ref, 1e-3, 'sigma at 256^3 (strided subset)')


def test_renderer_falls_back_for_uncompiled_decoder_width(gpu_device):
    """decoder_hidden = 48 has no fused kernel (IDE3D_ENOKERNEL): the renderer must take the step-wise HIP ops, not crash."""
    from training import triplane
    from training import volumetric_rendering as vr
    torch.manual_seed(3)
    sp = triplane.tiny_spec(decoder_hidden=48)
    R = triplane.TriplaneRenderer(sp).to(gpu_device).eval()
    g = torch.Generator().manual_seed(4)
    tex = torch.randn(2, 48, 32, 32, generator=g).to(gpu_device); geo = torch.randn(2, 48, 32, 32, generator=g).to(gpu_device)
    cam = torch.cat([triplane.camera_label(0.3), triplane.camera_label(-0.2)])[:, :16].reshape(-1, 4, 4).to(gpu_device)
    jit = torch.rand(2, 64, 12, generator=g).to(gpu_device)
    vr._init()
    assert vr.render_triplane_fused(tex.contiguous(memory_format=torch.channels_last), geo.contiguous(memory_format=torch.channels_last),
                                    R.decoder.kernel_weights(), cam, sp.fov, (8, 8), 12, sp.ray_start, sp.ray_end, jitter=jit) is None
    b_rr, b_c, b_g = _calls('render_rays'), _calls('composite'), _calls('triplane_sample_rays') + _calls('triplane_sample')
    with torch.no_grad():
        feat, depth, wsum = R(tex, geo, cam, jitter=jit)
    assert _calls('render_rays') == b_rr and _calls('composite') == b_c + 1
    assert _calls('triplane_sample_rays') + _calls('triplane_sample') == b_g + 2
    sd = {'synthesis.renderer.' + k: v.detach().cpu() for k, v in R.state_dict().items()}
    want = ogen.render(sd, ospec.tiny(decoder_hidden=48), tex.cpu(), geo.cpu(), cam.cpu(), jitter=jit.cpu(), ops=fast_ops)
    _rel(feat, want[0], 3e-4, 'step-wise features'); _rel(depth, want[1], 1e-4, 'step-wise depth'); _rel(wsum, want[2], 1e-4, 'weight sum')
    # mismatched decoder tensors are rejected before any launch (hip_plugin._fill_render_params)
    bad = dict(R.decoder.kernel_weights()); bad['tex_w0'] = bad['tex_w0'][:32].contiguous()
    with pytest.raises(RuntimeError, match='tex_w0'):
        vr._plugin.sample_voxel(tex.contiguous(memory_format=torch.channels_last), geo.contiguous(memory_format=torch.channels_last),
                                bad, torch.zeros(2, 4, 3, device=gpu_device))


def test_bench_parity_fixture_per_conv_arithmetic(bench_generator, gpu_device):
    """The frames `bench.py` counts, in every arithmetic of the 3x3 convolutions, against the oracle fixture the bench line itself
    uses (`parity`, tests/golden/bench_parity.npz): fp32 MFMA, bf16x6 and f16x3 agree with the CPU oracle to fp32 rounding
    (stated tolerance 2e-5 of the value range, measured 3e-6); bf16x3 to 1e-4 (measured 1.4e-5)."""
    import bench
    from torch_utils import hip_plugin
    from training import triplane
    G, _sd = bench_generator
    errs = {}
    try:
        for arith, tol in (('fp32', 2e-5), ('bf16x6', 2e-5), ('f16x3', 2e-5), ('bf16x3', 1e-4)):
            hip_plugin.conv_arithmetic(arith)
            before = hip_plugin.CALLS.get('modconv2d', 0)
            run = triplane.GraphedRenderer(G, 4, gpu_device)
            rec, _ = bench.check_parity(lambda z, c_cond, c_cam, jitter: run(z, c_cond, c_cam, jitter=jitter), gpu_device, tol=tol)
            assert rec['ok'], f'{arith}: {rec}'
            errs[arith] = max(rec['max_rel_err'].values())
    finally:
        hip_plugin.conv_arithmetic('default')
    assert errs['bf16x6'] < 2 * errs['fp32'] + 1e-6, f'bf16x6 is not fp32-grade at frame level: {errs}'
    assert errs['f16x3'] < 2 * errs['fp32'] + 1e-6, f'f16x3 is not fp32-grade at frame level: {errs}'
    assert errs['f16x3'] != errs['bf16x6'], 'f16x3 must actually run (every producer on the render path hands its amax on)'


def test_zz_write_measured_parity():
    """Not a check: stores the max relative errors the tests above measured (gpurun_out/parity_measured.json -> profiles/)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(root, 'gpurun_out', 'parity_measured.json'), 'w') as f:
        json.dump({k: MEASURED[k] for k in sorted(MEASURED)}, f, indent=1)
