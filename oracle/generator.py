"""CPU ORACLE of the generator forward pass — TEST / BASELINE INFRASTRUCTURE ONLY (rules: header of oracle/ops.py).

Functional restatement of `G.mapping` and `G.synthesis` for the tri-plane generator
on a plain `state_dict`: no nn.Module, no product code.  Layer semantics restate the reference blocks:
  mapping            inversion/networks.py:287-325 (normalize_2nd_moment :39, FullyConnectedLayer :152-165)
  synthesis layer    inversion/networks.py:420-514 (affine :432, fused modulated conv :82-130, noise :451-454,
                     bias_act :510-512); up-sampling conv = conv_transpose2d(stride 2) + 4x4 FIR, gain 4
                     (torch_utils/ops/conv2d_resample.py:112-129)
  toRGB / toSeg      inversion/networks.py:700-707
  dual-path block    inversion/networks.py:1053-1139
  ws slicing         extract_shapes.py:113-124
  renderer           training/volumetric_rendering.py:77-136, 34-74; dnnlib/util.py:580-617
The generator class itself is absent upstream; its topology here must equal `ide-3d_amd/training/triplane.py`
(same parameter names).  `make_golden.py` proves this restatement equal to the same topology assembled from the
REFERENCE's own modules (tests/golden/generator_tiny.npz).

`ops` selects the arithmetic: `oracle.ops` (float64 numpy, slow, tight) or `oracle.fast_ops` (fp32 torch CPU).
"""

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops as precise_ops

RESAMPLE = [1, 3, 3, 1]


def _fc(sd, prefix, x, lr_multiplier=1.0, act='linear', ops=precise_ops):
    w = sd[prefix + '.weight'].float() * (lr_multiplier / math.sqrt(sd[prefix + '.weight'].shape[1]))
    b = sd.get(prefix + '.bias')
    if b is not None:
        b = b.float() * lr_multiplier
    y = x @ w.t()
    if act == 'linear':
        return y + b if b is not None else y
    return ops.bias_act(y, b, act=act)


def _norm2(x, eps=1e-8):
    return x * (x.square().mean(dim=1, keepdim=True) + eps).rsqrt()


def mapping(sd, spec, z, c, truncation_psi=1, truncation_cutoff=None, ops=precise_ops):
    x = _norm2(z.float())
    y = _norm2(_fc(sd, 'mapping.embed', c.float(), ops=ops))
    x = torch.cat([x, y], dim=1)
    for i in range(spec.mapping_layers):
        x = _fc(sd, f'mapping.fc{i}', x, lr_multiplier=0.01, act='lrelu', ops=ops)
    ws = x.unsqueeze(1).repeat(1, num_ws(spec), 1)
    if truncation_psi != 1:
        w_avg = sd['mapping.w_avg'].float()
        if truncation_cutoff is None:
            ws = w_avg.lerp(ws, truncation_psi)
        else:
            ws[:, :truncation_cutoff] = w_avg.lerp(ws[:, :truncation_cutoff], truncation_psi)
    return ws


def num_ws(spec):
    n_vox = len(spec.voxel_resolutions())
    return (1 + 2 * (n_vox - 1)) + 2 * len(spec.sr_resolutions()) + 1


def _modconv(x, weight, styles, demodulate, up, ops):
    """Per-sample modulated (and demodulated) convolution, one image at a time."""
    n = x.shape[0]
    outs = []
    f = ops.setup_filter(RESAMPLE)
    for i in range(n):
        w = weight.float() * styles[i].reshape(1, -1, 1, 1)
        if demodulate:
            w = w * (w.square().sum(dim=[1, 2, 3], keepdim=True) + 1e-8).rsqrt()
        xi = x[i:i + 1]
        if up == 1:
            outs.append(F.conv2d(xi, w, padding=w.shape[-1] // 2))
        else:
            y = F.conv_transpose2d(xi, w.transpose(0, 1), stride=2)
            outs.append(ops.upfirdn2d(y, f, padding=[1, 1, 1, 1], gain=4))
    return torch.cat(outs, 0)


def _synthesis_layer(sd, prefix, x, w, up, noise_mode, conv_clamp, ops):
    styles = _fc(sd, prefix + '.affine', w, ops=ops)
    y = _modconv(x, sd[prefix + '.weight'], styles, True, up, ops)
    if noise_mode == 'const':
        y = y + sd[prefix + '.noise_const'].float() * sd[prefix + '.noise_strength'].float()
    elif noise_mode != 'none':
        raise ValueError('the oracle supports noise_mode const / none')
    return ops.bias_act(y, sd[prefix + '.bias'].float(), act='lrelu', clamp=conv_clamp)


def _to_head(sd, prefix, x, w, conv_clamp, ops):
    weight = sd[prefix + '.weight']
    styles = _fc(sd, prefix + '.affine', w, ops=ops) * (1 / math.sqrt(weight.shape[1] * weight.shape[2] ** 2))
    y = _modconv(x, weight, styles, False, 1, ops)
    return ops.bias_act(y, sd[prefix + '.bias'].float(), clamp=conv_clamp)


def _block(sd, prefix, x, img, seg, ws, first, noise_mode, conv_clamp, ops):
    """Dual-path block; ws [N, num_conv + 1, w_dim]."""
    n = ws.shape[0]
    k = 0
    if first:
        x = sd[prefix + '.const'].float().unsqueeze(0).expand(n, -1, -1, -1)
    else:
        x = _synthesis_layer(sd, prefix + '.conv0', x, ws[:, k], 2, noise_mode, conv_clamp, ops); k += 1
    x = _synthesis_layer(sd, prefix + '.conv1', x, ws[:, k], 1, noise_mode, conv_clamp, ops); k += 1
    f = ops.setup_filter(RESAMPLE)
    w_shared = ws[:, k]
    if img is not None and img.shape[-1] * 2 == x.shape[-1]:
        img = ops.upsample2d(img, f)
    if seg is not None and seg.shape[-1] * 2 == x.shape[-1]:
        seg = ops.upsample2d(seg, f)
    y = _to_head(sd, prefix + '.torgb', x, w_shared, conv_clamp, ops)
    img = y if img is None else img + y
    y = _to_head(sd, prefix + '.toseg', x, w_shared, conv_clamp, ops)
    seg = y if seg is None else seg + y
    return x, img, seg


def split_ws(spec, ws):
    voxel, sr, idx = [], [], 0
    for i, _res in enumerate(spec.voxel_resolutions()):
        nconv = 1 if i == 0 else 2
        voxel.append(ws[:, idx:idx + nconv + 1]); idx += nconv
    for _res in spec.sr_resolutions():
        sr.append(ws[:, idx:idx + 3]); idx += 2
    return voxel, sr


def backbone(sd, spec, ws, noise_mode='const', ops=precise_ops):
    voxel_ws, _ = split_ws(spec, ws.float())
    x = img = seg = None
    for i, res in enumerate(spec.voxel_resolutions()):
        x, img, seg = _block(sd, f'synthesis.vb{res}', x, img, seg, voxel_ws[i], i == 0, noise_mode, spec.conv_clamp, ops)
    return img, seg


def sample_voxel(sd, spec, img_v, seg_v, pts, ops=precise_ops):
    """[B, M, 3] -> [B*M, feat + seg + 1] (sigma last)."""
    geo = ops.sample_from_triplane(pts, seg_v)
    tex = ops.sample_from_triplane(pts, img_v)
    d = 'synthesis.renderer.decoder.'
    g = _fc(sd, d + 'geo1', _fc(sd, d + 'geo0', geo, act='softplus', ops=ops), ops=ops)
    t = _fc(sd, d + 'tex1', _fc(sd, d + 'tex0', tex, act='softplus', ops=ops), ops=ops)
    return torch.cat([t, g[:, 1:], g[:, :1]], dim=1)


def render(sd, spec, img_v, seg_v, cam2world, jitter=None, sigma_noise=None, ops=precise_ops, num_steps=None,
           hierarchical=False, importance_u=None):
    """-> features [N, feat+seg, R, R], depth [N, 1, R, R], weight sum [N, 1, R, R].
    hierarchical: SURVEY.md 3.5 step 6 — the first pass's weights (+1e-5, end samples dropped) define a pdf over the
    depth mid-points, `sample_pdf` draws `steps` more depths per ray with `importance_u` [N*R*R, steps], those points
    are queried, both sets are merged by depth and integrated together."""
    n = img_v.shape[0]
    size = spec.render_size
    steps = spec.num_steps if num_steps is None else num_steps
    points, z_vals, d_cam = ops.initial_rays(n, steps, spec.fov, (size, size), spec.ray_start, spec.ray_end)
    if jitter is not None:
        points, z_vals = ops.perturb(points, z_vals, d_cam, jitter.reshape(n, size * size, steps, 1))
    cam2world = cam2world.float()
    world = ops.to_world(points, cam2world)
    out = sample_voxel(sd, spec, img_v, seg_v, world.reshape(n, -1, 3), ops).reshape(n, size * size, steps, -1)
    noise = None if sigma_noise is None else sigma_noise.reshape(n, size * size, steps, 1)
    if hierarchical:
        rays = size * size
        _, _, w = ops.composite(out, d_cam, z_vals, noise=noise, clamp_mode=spec.clamp_mode)
        w = w.reshape(n * rays, steps) + 1e-5
        z = z_vals.reshape(n * rays, steps)
        z_fine = ops.sample_pdf(0.5 * (z[:, :-1] + z[:, 1:]), w[:, 1:-1], importance_u).reshape(n, rays, steps, 1)
        rot_only = cam2world.clone()
        rot_only[:, :3, 3] = 0
        dirs = ops.to_world(d_cam, rot_only)                                   # [n, rays, 3]
        origin = cam2world[:, None, None, :3, 3]
        fine = origin + dirs[:, :, None, :] * z_fine
        out_f = sample_voxel(sd, spec, img_v, seg_v, fine.reshape(n, -1, 3), ops).reshape(n, rays, steps, -1)
        z_vals, order = torch.sort(torch.cat([z_fine, z_vals], 2), dim=2)
        out = torch.gather(torch.cat([out_f, out], 2), 2, order.expand(-1, -1, -1, out.shape[-1]))
        noise = None
    feat, depth, weights = ops.composite(out, d_cam, z_vals, noise=noise, clamp_mode=spec.clamp_mode)
    feat = feat.permute(0, 2, 1).reshape(n, -1, size, size)
    depth = depth.permute(0, 2, 1).reshape(n, 1, size, size)
    wsum = weights.sum(2).permute(0, 2, 1).reshape(n, 1, size, size)
    return feat, depth, wsum


def superres(sd, spec, feat, ws, noise_mode='const', ops=precise_ops):
    _, sr_ws = split_ws(spec, ws.float())
    fc = spec.feature_channels
    size = spec.sr_resolutions()[0] // 2
    up = lambda t: F.interpolate(t, size=(size, size), mode='bilinear', align_corners=False)
    x, img, seg = up(feat[:, :fc]), up(feat[:, :spec.img_channels]), up(feat[:, fc:])
    for i, res in enumerate(spec.sr_resolutions()):
        x, img, seg = _block(sd, f'synthesis.b{res}', x, img, seg, sr_ws[i], False, noise_mode, spec.conv_clamp, ops)
    return img, seg


def synthesis(sd, spec, ws, c, jitter=None, noise_mode='const', ops=precise_ops):
    sd = {k: v.detach().cpu() for k, v in sd.items()}
    with torch.no_grad():
        img_v, seg_v = backbone(sd, spec, ws, noise_mode, ops)
        cam2world = c[:, :16].reshape(-1, 4, 4).float()
        feat, depth, wsum = render(sd, spec, img_v, seg_v, cam2world, jitter=jitter, ops=ops)
        img, seg = superres(sd, spec, feat, ws, noise_mode, ops)
    return dict(image=img, image_seg=seg, image_raw=feat[:, :spec.img_channels], image_depth=depth,
                features=feat, planes=(img_v, seg_v))
