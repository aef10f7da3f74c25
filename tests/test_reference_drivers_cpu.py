"""The UNTOUCHED reference driver code on top of the overlay (INTEGRATION.md option A), in this container on CPU tensors:

  * `scripts/make_random_init_pkl.py` writes `dict(G_ema=G)`; the REFERENCE's `legacy.load_network_pkl` (legacy.py:22-42) loads it
    with `sys.path = [overlay, reference]`, i.e. `torch_utils` / `training` / `dnnlib` resolve to this repository and everything the
    overlay does not carry (`legacy`, `dnnlib.seg_tools`, ...) to the reference checkout;
  * the call sequence of gen_images.py:85-111 runs verbatim on the loaded generator (`G.mapping(z=, c=, truncation_psi=)`,
    `sample_camera_positions`, `create_cam2world_matrix`, `G.synthesis(ws, c=c, render_params=..., noise_mode=..., return_seg=True)`,
    the reference's `mask2color`), and equals the product called directly;
  * the viewer pattern of viz/renderer.py:410-441 (a forward hook on every sub-module, `force_fp32=` keyword) works on the module tree.

Needs /root/reference (absent on the GPU box: skipped there).  Modules the image lacks and the drivers import at module level
(cv2, torchvision, numpy-1 private modules) are stubbed exactly as oracle/ref_import.py does for the golden generation.
"""

import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('IDE3D_REFERENCE', '/root/reference')

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'torch_utils')), reason='reference checkout not present')

CHILD = r'''
import math, os, sys, types
overlay, ref, pkl = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path[:0] = [overlay, ref]
import numpy as np, numpy.lib
# stubs for modules this image lacks (same set as oracle/ref_import.py, plus torchvision for dnnlib/seg_tools.py:6)
sys.modules.setdefault('cv2', types.ModuleType('cv2'))
for name, attrs in (('arraysetops', dict(isin=np.isin)), ('function_base', dict(angle=np.angle, iterable=np.iterable))):
    m = types.ModuleType('numpy.lib.' + name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules['numpy.lib.' + name] = m; setattr(numpy.lib, name, m)
tv = types.ModuleType('torchvision'); tv.transforms = types.ModuleType('torchvision.transforms'); tv.utils = types.ModuleType('torchvision.utils')
sys.modules.update({'torchvision': tv, 'torchvision.transforms': tv.transforms, 'torchvision.utils': tv.utils})

import torch
import dnnlib, legacy                                     # legacy.py is the reference's file; dnnlib the overlay (extended over the reference)
from training.volumetric_rendering import sample_camera_positions, create_cam2world_matrix     # gen_images.py:12
from dnnlib.seg_tools import *                            # gen_images.py:14 -> the REFERENCE's module through the extended package path
assert legacy.__file__.startswith(ref) and dnnlib.__file__.startswith(overlay)
assert sys.modules['dnnlib.seg_tools'].__file__.startswith(ref)
# dnnlib/seg_tools.py:10 `from inversion.BiSeNet import BiSeNet` now finds the overlay's face parser; the rest of `inversion` stays the reference's
import inversion.BiSeNet, inversion.networks
assert inversion.BiSeNet.__file__.startswith(overlay) and inversion.networks.__file__.startswith(ref)
assert sys.modules['dnnlib.seg_tools'].BiSeNet.__module__ == 'training.face_parsing'
_net = BiSeNet(n_classes=20).eval()
_lab = face_parsing(torch.zeros(1, 3, 64, 64), _net)          # the REFERENCE's face_parsing / parsing_img / id_remap / scatter (seg_tools.py:100-123) on it
assert _lab.shape == (1, 19, 512, 512) and float(_lab.sum(1).min()) == 1.0
import training.networks, training.triplane, torch_utils.ops.upfirdn2d
for mod in (training.networks, training.triplane, torch_utils.ops.upfirdn2d, sys.modules['training.volumetric_rendering']):
    assert mod.__file__.startswith(overlay), mod.__file__

device = torch.device('cpu')
with dnnlib.util.open_url(pkl) as f:                      # gen_images.py:81-82
    G = legacy.load_network_pkl(f)['G_ema'].to(device)
assert type(G).__module__ == 'training.triplane'

truncation_psi, noise_mode = 0.7, 'const'
cs = torch.tensor([1,0,0,0, 0,1,0,0, 0,0,1,2.7, 0,0,0,1, 4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).float().to(device).reshape(1,-1)
outs = []
for seed in (0, 1):
    torch.manual_seed(seed)
    z = torch.from_numpy(np.random.RandomState(seed).randn(1, G.z_dim)).to(device)
    ws = G.mapping(z=z, c=cs, truncation_psi=truncation_psi)
    for k, yaw in enumerate([-0.5, 0, 0.5]):
        render_params = {"h_mean": yaw+math.pi*0.5, "v_mean": math.pi * 0.5, "h_stddev": 0., "v_stddev": 0., "fov": 18, "num_steps": 12}
        camera_points, phi, theta = sample_camera_positions(device, n=1, r=2.7, horizontal_mean=yaw+math.pi*0.5, vertical_mean=math.pi * 0.5, mode=None)
        c = create_cam2world_matrix(-camera_points, camera_points, device=device)
        c = c.reshape(1,-1)
        c = torch.cat((c, torch.tensor([4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).reshape(1, -1).to(c)), -1)
        torch.manual_seed(1000 + k)                       # the stratified jitter is drawn inside synthesis
        img, seg = G.synthesis(ws, c=c, render_params=render_params, noise_mode=noise_mode, return_seg=True)
        col = (mask2color(seg) / 255. - 0.5) / 0.5
        assert img.shape == (1, 3, G.img_resolution, G.img_resolution) and col.shape == img.shape
        # the product called directly with the same draws
        from training import triplane
        torch.manual_seed(1000 + k)
        img2, seg2 = G.synthesis(ws, c=triplane.camera_label(yaw), noise_mode='const', return_seg=True)
        assert torch.allclose(img, img2, atol=1e-5) and torch.allclose(seg, seg2, atol=1e-5)
        outs.append(float(img.abs().mean()))
assert all(math.isfinite(v) and v > 0 for v in outs)

# viz/renderer.py:410-441 pattern: hooks on every sub-module, force_fp32 keyword, dict output
names = {mod: name for name, mod in G.named_modules()}
seen = []
hooks = [m.register_forward_hook(lambda mod, _i, out: seen.append(names[mod])) for m in G.modules()]
out = G.synthesis(ws, c=c, noise_mode='const', force_fp32=True, return_dict=True)
for h in hooks:
    h.remove()
assert set(out) >= {'image', 'image_depth', 'image_raw'}
assert any(n.startswith('synthesis.vb') for n in seen) and any(n.startswith('synthesis.b') and 'torgb' in n for n in seen) and 'synthesis.renderer' in seen
print('DRIVER_SMOKE_OK', len(seen))
'''


def test_untouched_reference_driver_sequence(tmp_path):
    pkl = tmp_path / 'random-init-tiny.pkl'
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'make_random_init_pkl.py'), '--tiny', '--out', str(pkl)],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    child = tmp_path / 'child.py'
    child.write_text(textwrap.dedent(CHILD))
    env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}
    res = subprocess.run([sys.executable, str(child), os.path.join(ROOT, 'ide-3d_amd'), REF, str(pkl)], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0 and 'DRIVER_SMOKE_OK' in res.stdout, res.stdout[-4000:]


# ---- gen_videos.py / extract_shapes.py: the reference's own loops, imported as modules, on the overlay -----------------------------
CHILD_LOOPS = r'''
import math, os, sys, types
overlay, ref, pkl = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path[:0] = [overlay, ref]
import numpy as np, numpy.lib
# modules this image lacks and the two drivers import at module level (nothing of them is on the tested path, except the video writer,
# whose stub records the frames it is given)
cv2 = types.ModuleType('cv2'); cv2.normalize = None; sys.modules['cv2'] = cv2
for name, attrs in (('arraysetops', dict(isin=np.isin)), ('function_base', dict(angle=np.angle, iterable=np.iterable))):
    m = types.ModuleType('numpy.lib.' + name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules['numpy.lib.' + name] = m; setattr(numpy.lib, name, m)
tv = types.ModuleType('torchvision'); tv.transforms = types.ModuleType('torchvision.transforms'); tv.utils = types.ModuleType('torchvision.utils')
tv.utils.save_image = None
sys.modules.update({'torchvision': tv, 'torchvision.transforms': tv.transforms, 'torchvision.utils': tv.utils})
for name in ('plyfile', 'mrcfile', 'skimage', 'skimage.measure'):
    sys.modules[name] = types.ModuleType(name)
sys.modules['skimage'].measure = sys.modules['skimage.measure']
written = {}
class _Writer:
    def __init__(self, path, **kw): self.path, self.kw, self.frames = path, kw, []; written[path] = self
    def append_data(self, frame): self.frames.append(np.array(frame))
    def close(self): self.closed = True
imageio = types.ModuleType('imageio'); imageio.get_writer = lambda path, **kw: _Writer(path, **kw); sys.modules['imageio'] = imageio

import torch
import dnnlib, legacy
import gen_videos, extract_shapes                          # the REFERENCE's files
assert gen_videos.__file__.startswith(ref) and extract_shapes.__file__.startswith(ref)
import training.video_render as video_render, training.shape_extraction as shape_extraction
assert video_render.__file__.startswith(overlay)

device = torch.device('cpu')
with dnnlib.util.open_url(pkl) as f:
    G = legacy.load_network_pkl(f)['G_ema'].to(device)
# neither driver can pass jitter draws: render without stratified jitter on both sides (keyword injected in front of G.synthesis)
G.synthesis.register_forward_pre_hook(lambda mod, args, kwargs: (args, {**kwargs, 'ray_jitter': False}), with_kwargs=True)

# gen_videos.py:66-139, 2 x 2 grid of four seeds, image_seg frames, 2 frames
seeds, w_frames = [0, 1, 2, 3], 2
gen_videos.gen_interp_video(G=G, mp4='grid.mp4', seeds=seeds, w_frames=w_frames, grid_dims=(2, 2), image_mode='image_seg', device=device)
wr = written['grid.mp4']
assert wr.kw.get('codec') == 'libx264' and wr.kw.get('fps') == 60 and getattr(wr, 'closed', False)     # gen_videos.py:108,139
ref_frames = np.stack(wr.frames)
res = G.img_resolution
assert ref_frames.shape == (w_frames, 2 * res, 2 * 2 * res, 3) and ref_frames.dtype == np.uint8
ours = np.stack([f.cpu().numpy() for f in video_render.gen_interp_frames(G, seeds, w_frames=w_frames, grid_dims=(2, 2), image_mode='image_seg', device=device)])
assert ours.shape == ref_frames.shape
# the product renders the four cells of a frame in one batch from cached tri-planes; the reference one by one from scratch: identical
# up to fp32 summation order inside ATen's CPU convolutions, i.e. isolated uint8 roundings / argmax flips
diff = ours.astype(np.int32) - ref_frames.astype(np.int32)
rgb = np.concatenate([diff[:, :, :res], diff[:, :, 2 * res:3 * res]], axis=2)            # the RGB halves of the two grid columns
assert np.abs(rgb).max() <= 2 and (diff != 0).mean() < 5e-3, (int(np.abs(rgb).max()), float((diff != 0).mean()))

# extract_shapes.py:99-150 at voxel_resolution 32 (driver arguments of extract_shapes.py:166-189)
label = torch.tensor([1,0,0,0, 0,1,0,0, 0,0,1,2.7, 0,0,0,1, 4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1])[None].float()
render_params = {"h_stddev": 0., "v_stddev": 0., "num_steps": 96}
z = torch.from_numpy(np.random.RandomState(5).randn(1, G.z_dim))
cube_ref = extract_shapes.sample_generator_ide3d(G, None, z, label, cube_length=0.3, voxel_resolution=32, psi=1.0, max_batch=10000, **render_params)
cube = shape_extraction.sample_generator_ide3d(G, None, z, label, cube_length=0.3, voxel_resolution=32, psi=1.0, max_batch=10000, **render_params)
assert cube_ref.shape == (32, 32, 32) and cube.shape == cube_ref.shape
assert np.allclose(cube, cube_ref, rtol=1e-4, atol=1e-5 * np.abs(cube_ref).max()), float(np.abs(cube - cube_ref).max())
# the lattice the reference builds == the product's
s_ref, o_ref, v_ref = extract_shapes.create_samples(32, [0, 0, 0], 0.3)
s_our, o_our, v_our = shape_extraction.create_samples(32, [0, 0, 0], 0.3)
assert torch.equal(s_ref, s_our) and np.array_equal(o_ref, o_our) and v_ref == v_our
print('DRIVER_LOOPS_OK', ref_frames.shape, float(np.abs(cube_ref).max()))
'''


def test_reference_gen_videos_and_extract_shapes_loops(tmp_path):
    """gen_videos.py:66-139 (2 frames of a 2 x 2 `image_seg` grid) and extract_shapes.py:99-150 (32^3 cube) run from the REFERENCE's
    files on the overlay, with imageio / mrcfile / skimage / plyfile / cv2 / torchvision stubbed; results equal
    training/video_render.py / training/shape_extraction.py called directly."""
    pkl = tmp_path / 'random-init-tiny.pkl'
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'make_random_init_pkl.py'), '--tiny52', '--out', str(pkl)],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    child = tmp_path / 'child_loops.py'
    child.write_text(textwrap.dedent(CHILD_LOOPS))
    env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}
    res = subprocess.run([sys.executable, str(child), os.path.join(ROOT, 'ide-3d_amd'), REF, str(pkl)], env=env, cwd=str(tmp_path),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert res.returncode == 0 and 'DRIVER_LOOPS_OK' in res.stdout, res.stdout[-4000:]
