"""CPU ORACLE (fast variant) — TEST / BASELINE INFRASTRUCTURE ONLY (see the header of oracle/ops.py for the rules).

fp32 multi-threaded PyTorch-CPU formulations of the same ops as `oracle/ops.py`, following how the reference's own
CPU path evaluates them (`impl='ref'`: pad + depth-wise `conv2d` for upfirdn2d, `grid_sample` for the tri-plane,
`cumprod` for compositing).  Used (a) as the `cpu_baseline` "port" that `bench.py` times on the GPU box's host
cores, where /root/reference does not exist, and (b) to run the full-size generator oracle in seconds instead of
minutes.  Checked against the same golden vectors as `oracle/ops.py` (tests/test_oracle_golden.py).
"""

import math

import torch
import torch.nn.functional as F

from . import ops as _precise

setup_filter = _precise.setup_filter
camera_rays = _precise.camera_rays
initial_rays = _precise.initial_rays
perturb = _precise.perturb
cam2world_lookat = _precise.cam2world_lookat
camera_position = _precise.camera_position
lookat_pose = _precise.lookat_pose
create_samples = _precise.create_samples
frame_u8 = _precise.frame_u8
sample_pdf = _precise.sample_pdf

_ACTS = {
    'linear': lambda x, a: x,
    'relu': lambda x, a: F.relu(x),
    'lrelu': lambda x, a: F.leaky_relu(x, a),
    'tanh': lambda x, a: torch.tanh(x),
    'sigmoid': lambda x, a: torch.sigmoid(x),
    'elu': lambda x, a: F.elu(x),
    'selu': lambda x, a: F.selu(x),
    'softplus': lambda x, a: F.softplus(x),
    'swish': lambda x, a: torch.sigmoid(x) * x,
}


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """bias_act.py:91-120."""
    def_alpha, def_gain = _precise._ACT_DEFAULTS[act]
    alpha = def_alpha if alpha is None else float(alpha)
    gain = def_gain if gain is None else float(gain)
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    x = _ACTS[act](x, alpha)
    if gain != 1:
        x = x * gain
    if clamp is not None and clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.py:167-211 (zero-stuff, pad / crop, depth-wise correlation with the flipped filter, decimate)."""
    n, c, ih, iw = x.shape
    ux, uy = _precise._pair(up)
    dx, dy = _precise._pair(down)
    px0, px1, py0, py1 = _precise._pad4(padding)
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    u = x.new_zeros([n, c, ih, uy, iw, ux])
    u[:, :, :, 0, :, 0] = x
    u = u.reshape(n, c, ih * uy, iw * ux)
    u = F.pad(u, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    u = u[:, :, max(-py0, 0): u.shape[2] - max(-py1, 0), max(-px0, 0): u.shape[3] - max(-px1, 0)]
    k = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:
        k = k.flip(list(range(k.ndim)))
    if k.ndim == 2:
        u = F.conv2d(u, k[None, None].repeat(c, 1, 1, 1), groups=c)
    else:
        u = F.conv2d(u, k[None, None, None, :].repeat(c, 1, 1, 1), groups=c)
        u = F.conv2d(u, k[None, None, :, None].repeat(c, 1, 1, 1), groups=c)
    return u[:, :, ::dy, ::dx]


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1):
    ux, uy = _precise._pair(up)
    px0, px1, py0, py1 = _precise._pad4(padding)
    fw, fh = f.shape[-1], f.shape[0]
    p = [px0 + (fw + ux - 1) // 2, px1 + (fw - ux) // 2, py0 + (fh + uy - 1) // 2, py1 + (fh - uy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * ux * uy)


def sample_from_triplane(coordinates, grid):
    """dnnlib/util.py:580-617 on ATen grid_sample."""
    n, c3, h, w = grid.shape
    planes = grid.reshape(n, 3, c3 // 3, h, w)
    out = 0
    for pl, axes in enumerate(([0, 1], [1, 2], [0, 2])):
        g = coordinates[..., axes].reshape(n, -1, 1, 2)
        s = F.grid_sample(planes[:, pl], g, mode='bilinear', padding_mode='zeros', align_corners=False)
        nn_, cc, hh, ww = s.shape
        out = out + s.permute(0, 3, 2, 1).reshape(nn_ * hh * ww, cc)
    return out


def to_world(points, cam2world):
    """volumetric_rendering.py:123-127."""
    n = points.shape[0]
    homo = torch.ones(points.shape[:-1] + (4,))
    homo[..., :3] = points
    out = torch.bmm(cam2world, homo.reshape(n, -1, 4).permute(0, 2, 1)).permute(0, 2, 1)
    return out.reshape(points.shape[:-1] + (4,))[..., :3]


def composite(rgb_sigma, rays_d_cam, z_vals, noise=None, last_back=False, white_back=False, max_depth=None,
              clamp_mode='softplus', fill_mode=None):
    """volumetric_rendering.py:34-74."""
    rgbs, sigmas = rgb_sigma[..., :-1], rgb_sigma[..., -1:]
    deltas = (z_vals[:, :, 1:] - z_vals[:, :, :-1]) * torch.norm(rays_d_cam, p=2, dim=-1, keepdim=True).unsqueeze(2)
    deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :, :1])], -2)
    if noise is not None:
        sigmas = sigmas + noise
    dens = F.softplus(sigmas) if clamp_mode == 'softplus' else F.relu(sigmas)
    alphas = 1 - torch.exp(-deltas * dens)
    shifted = torch.cat([torch.ones_like(alphas[:, :, :1]), 1 - alphas + 1e-10], -2)
    weights = alphas * torch.cumprod(shifted, -2)[:, :, :-1]
    wsum = weights.sum(2)
    if last_back:
        weights[:, :, -1] += (1 - wsum)
    rgb = torch.sum(weights * rgbs, -2)
    depth = torch.sum(weights * z_vals, -2)
    if white_back:
        rgb = rgb + 1 - wsum
    if max_depth:
        depth = depth + (1 - wsum) * max_depth
    if fill_mode == 'debug':
        rgb[wsum.squeeze(-1) < 0.9] = torch.tensor([1., 0, 0])
    elif fill_mode == 'weight':
        rgb = wsum.expand_as(rgb)
    return rgb, depth, weights
