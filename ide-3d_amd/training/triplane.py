"""`TriPlaneGenerator` — the IDE-3D "64-neural-render -> 512" generator (`training.triplane` is the module
the reference's viewer imports it from, viz/renderer.py:196).

The class itself is NOT in the reference repository: it ships inside the released pickle
(SURVEY.md §0.1).  This module re-specifies it from the call-site evidence collected in SURVEY.md §3.5 /
Appendix B, out of the reference's own building blocks:

    ws = G.mapping(z, c)                                   gen_images.py:92
    img, seg = G.synthesis(ws, c=c, render_params=..., noise_mode=..., return_seg=True)   gen_images.py:109
    G.synthesis.voxel_block_resolutions / vb{res}(x, img, ws, condition_img=seg) -> (x, img, seg)
    G.synthesis.block_resolutions / b{res},  G.synthesis.render_size,  G.synthesis.num_ws / w_dim
    G.synthesis.renderer.sample_voxel(img_v, seg_v, pts) -> [B*M, 52]  (sigma last)        extract_shapes.py:113-147

Data flow of `synthesis` (N images):
    vb4 .. vb256 (dual-path StyleGAN2 blocks)  -> texture tri-plane img_v [N, 3*32, 256, 256]
                                                  semantic/geometry tri-plane seg_v [N, 3*32, 256, 256]
    renderer: 64x64 rays x 96 samples, both tri-planes gathered, decoded by two small MLPs to
              [32 colour features | 19 semantic logits | sigma], alpha-composited  -> [N, 51, 64, 64]
    b256, b512 (dual-path blocks) on the bilinearly 2x up-sampled 32-channel feature image with the raw RGB /
              semantic images as skip inputs -> img [N, 3, 512, 512], seg [N, 19, 512, 512]

On an MI355X the renderer stage is a single HIP launch (`training.volumetric_rendering.render_triplane_fused`).
Every architecture number is a constructor argument (`GeneratorSpec`): the real widths are unknown and all
parity statements are "this path vs the CPU oracle with the same spec and weights".
"""

import contextlib
import dataclasses
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from dnnlib import util
from torch_utils import misc
from torch_utils import persistence
from torch_utils.ops import bias_act
from training import graph_cache
from training import networks
from training import volumetric_rendering as vr


@dataclasses.dataclass
class GeneratorSpec:
    """Architecture of the random-init ide3d-ffhq-64-512 generator."""
    z_dim: int = 512
    c_dim: int = 25
    w_dim: int = 512
    img_resolution: int = 512
    img_channels: int = 3
    seg_channels: int = 19
    mapping_layers: int = 8
    channel_base: int = 32768
    channel_max: int = 512
    plane_resolution: int = 256
    plane_channels: int = 32          # per plane; a tri-plane tensor has 3x this
    render_size: int = 64
    feature_channels: int = 32        # colour feature width (first 3 = raw RGB)
    decoder_hidden: int = 64
    sr_channels: Optional[Dict[int, int]] = None     # SR block widths; default {256: 128, 512: 64}
    num_steps: int = 96
    ray_start: float = 2.25
    ray_end: float = 3.3
    fov: float = 18.0
    conv_clamp: Optional[float] = None
    num_fp16_res: int = 0             # StyleGAN2-style fp16 storage for the highest resolutions (0 = all fp32); implies conv_clamp 256
    clamp_mode: str = 'softplus'
    hierarchical: bool = False        # second, importance-sampled pass of `num_steps` more samples per ray

    def sr_resolutions(self) -> List[int]:
        # two 2x blocks end at img_resolution: e.g. 64 -> (bilinear 128) -> 256 -> 512
        return [self.img_resolution // 2, self.img_resolution]

    def sr_widths(self) -> Dict[int, int]:
        if self.sr_channels is not None:
            return dict(self.sr_channels)
        r0, r1 = self.sr_resolutions()
        return {r0: min(self.channel_base // r0, self.channel_max), r1: min(self.channel_base // r1, self.channel_max)}

    def voxel_resolutions(self) -> List[int]:
        return [2 ** i for i in range(2, int(np.log2(self.plane_resolution)) + 1)]

    def voxel_width(self, res: int) -> int:
        return min(self.channel_base // res, self.channel_max)


def tiny_spec(**overrides) -> GeneratorSpec:
    """A seconds-on-CPU configuration with the same topology (used by tests and golden fixtures)."""
    base = dict(z_dim=32, c_dim=25, w_dim=32, img_resolution=64, mapping_layers=2, channel_base=256, channel_max=16,
                plane_resolution=32, plane_channels=16, render_size=8, feature_channels=8, seg_channels=5,
                decoder_hidden=32, num_steps=12)
    base.update(overrides)
    return GeneratorSpec(**base)


@persistence.persistent_class
class VoxelBlock(networks.SegSynthesisBlock):
    """`vb{res}`: dual-path block with the call signature the reference drivers use
    (`block(x, img, ws, condition_img=seg) -> (x, img, seg)`, extract_shapes.py:126-129)."""

    def forward(self, x, img, ws, condition_img=None, **kwargs):
        return super().forward(x, img, condition_img, ws, **kwargs)


@persistence.persistent_class
class TriplaneDecoder(torch.nn.Module):
    """Two 2-layer MLPs: geometry/semantic branch (sigma + seg logits) on the semantic tri-plane feature,
    texture branch (colour features) on the texture tri-plane feature.  Hidden activation softplus."""

    def __init__(self, in_channels, hidden, feature_channels, seg_channels):
        super().__init__()
        self.in_channels, self.hidden = in_channels, hidden
        self.feature_channels, self.seg_channels = feature_channels, seg_channels
        self.geo0 = networks.FullyConnectedLayer(in_channels, hidden, activation='softplus')
        self.geo1 = networks.FullyConnectedLayer(hidden, 1 + seg_channels)
        self.tex0 = networks.FullyConnectedLayer(in_channels, hidden, activation='softplus')
        self.tex1 = networks.FullyConnectedLayer(hidden, feature_channels)

    def forward(self, tex_feat, geo_feat):
        """[M, C] features -> [M, feature + seg + 1] with sigma last."""
        g = self.geo1(self.geo0(geo_feat))
        t = self.tex1(self.tex0(tex_feat))
        return torch.cat([t, g[:, 1:], g[:, :1]], dim=1)

    def kernel_weights(self):
        """Effective (gain-folded) weights in the layout `ide3d_render_rays` expects."""
        out = {}
        for name, layer in (('geo_w0', self.geo0), ('geo_w1', self.geo1), ('tex_w0', self.tex0), ('tex_w1', self.tex1)):
            if layer.weight.is_cuda and layer.weight.dtype == torch.float32 and not (torch.is_grad_enabled() and layer.weight.requires_grad):
                # inference: the gain-folded copies are formed once per (tensor, version), not by four element-wise launches per call
                # (a density query of extract_shapes.py's chunk loop is ONE other launch)
                w = networks._scaled_weight(layer.weight, layer.weight_gain)
                b = layer.bias if layer.bias_gain == 1 else networks._scaled_weight(layer.bias, layer.bias_gain)
            else:
                w, b = layer.effective(torch.float32)
            out[name] = w.contiguous()
            out[name.replace('_w', '_b')] = b.contiguous()
        return out


@persistence.persistent_class
class TriplaneRenderer(torch.nn.Module):
    """`G.synthesis.renderer`: tri-plane sampling + decoding + volume integration."""

    def __init__(self, spec: GeneratorSpec):
        super().__init__()
        self.spec = spec
        self.decoder = TriplaneDecoder(spec.plane_channels, spec.decoder_hidden, spec.feature_channels, spec.seg_channels)

    # -- point queries ---------------------------------------------------------------------------------
    def sample_voxel(self, img_v, seg_v, pts, sigma_only=False, ray_grid=None):
        """Features at world points.  img_v / seg_v [B, 3C, H, W], pts [B, M, 3] -> [B*M, feat + seg + 1]
        (sigma last; extract_shapes.py:146).  `sigma_only=True` returns [B*M] densities.  `ray_grid=(H, W, steps)` tells the
        step-wise gather that pts is a flattened ray grid (see dnnlib.util.sample_from_triplane)."""
        if self._hip_ok(img_v, seg_v, pts):
            vr._init()
            tex, geo = _as_channels_last(img_v), _as_channels_last(seg_v)
            out = vr._plugin.sample_voxel(tex, geo, self.decoder.kernel_weights(), pts.float(), sigma_only=sigma_only)
            if out is not None:
                return out
        geo_feat = util.sample_from_triplane(pts, seg_v, ray_grid=ray_grid)
        tex_feat = util.sample_from_triplane(pts, img_v, ray_grid=ray_grid)
        out = self.decoder(tex_feat, geo_feat)
        return out[:, -1] if sigma_only else out

    def density_lattice(self, img_v, seg_v, n, voxel_size, corner, scale, first, count):
        """Densities of points [first, first+count) of the extract_shapes.py lattice (training.shape_extraction) -> [B*count].
        On the GPU the points are generated inside the fused kernel; elsewhere they are materialised and queried."""
        if self._hip_ok(img_v, seg_v):
            vr._init()
            out = vr._plugin.density_lattice(_as_channels_last(img_v), _as_channels_last(seg_v), self.decoder.kernel_weights(),
                                             n, voxel_size, corner, scale, first, count)
            if out is not None:
                return out
        from training import shape_extraction
        pts = shape_extraction.lattice_points(n, voxel_size, corner, scale, first, count, img_v.device)
        return self.sample_voxel(img_v, seg_v, pts.unsqueeze(0).expand(img_v.shape[0], -1, -1), sigma_only=True)

    @staticmethod
    def _hip_ok(*tensors):
        if any(t.device.type != 'cuda' or t.dtype != torch.float32 for t in tensors):
            return False
        return not (torch.is_grad_enabled() and any(t.requires_grad for t in tensors))

    # -- full rendering -----------------------------------------------------------------------------------
    def forward(self, img_v, seg_v, cam2world, fov=None, num_steps=None, ray_start=None, ray_end=None, img_size=None,
                nerf_noise=0.0, jitter=None, sigma_noise=None, white_back=False, clamp_mode=None,
                hierarchical=None, importance_u=None, sigma_noise_fine=None, **unused):
        """Render N views.  Returns (features [N, feat+seg, R, R], depth [N, 1, R, R], weight sum [N, 1, R, R]).

        jitter: None -> draw U[0,1) per sample on the device (the reference always jitters,
                volumetric_rendering.py:113); False -> no jitter; tensor [N, R*R, S] -> use these draws.
        hierarchical: add the importance pass — S more depths per ray drawn with `sample_pdf` from the first pass's
                weights, queried, merged by depth and integrated together (2S samples).  `importance_u` [N*R*R, S] supplies
                the draws (None -> torch.rand); `sigma_noise_fine` [N, R*R, 2S] the density noise of the merged integration.
        """
        sp = self.spec
        fov = sp.fov if fov is None else fov
        steps = sp.num_steps if num_steps is None else num_steps
        t0 = sp.ray_start if ray_start is None else ray_start
        t1 = sp.ray_end if ray_end is None else ray_end
        size = sp.render_size if img_size is None else img_size
        clamp_mode = sp.clamp_mode if clamp_mode is None else clamp_mode
        n, device = img_v.shape[0], img_v.device
        rays = size * size
        if jitter is None:
            jitter = torch.rand([n, rays, steps], device=device)
        elif jitter is False:
            jitter = None
        if sigma_noise is None and nerf_noise:
            sigma_noise = torch.randn([n, rays, steps], device=device) * nerf_noise

        hierarchical = sp.hierarchical if hierarchical is None else hierarchical
        if not hierarchical and self._hip_ok(img_v, seg_v, cam2world):
            res = vr.render_triplane_fused(_as_channels_last(img_v), _as_channels_last(seg_v), self.decoder.kernel_weights(),
                                           cam2world.float(), fov, (size, size), steps, t0, t1, jitter=jitter,
                                           sigma_noise=sigma_noise, clamp_mode=clamp_mode, white_back=white_back)
            if res is not None:
                return res

        # step-wise definition (CPU / autograd / configurations outside the fused kernel)
        points, z_vals, rays_d_cam = vr.get_initial_rays_trig(n, steps, device, fov, (size, size), t0, t1)
        pts, z_vals, rays_d, rays_o, _p, _y = vr.transform_sampled_points(
            points, z_vals, rays_d_cam, device, h_stddev=0, v_stddev=0, camera=cam2world, mode=None,
            jitter=(jitter.unsqueeze(-1) if jitter is not None else torch.full_like(z_vals, 0.5)))
        out = self.sample_voxel(img_v, seg_v, pts.reshape(n, -1, 3), ray_grid=(size, size, steps)).reshape(n, rays, steps, -1)
        noise = sigma_noise.unsqueeze(-1) if sigma_noise is not None else None
        integrate = lambda o, z, nz: vr.fancy_integration(o, rays_d_cam, z, device, noise_std=(1.0 if nz is not None else 0.0),
                                                          white_back=white_back, clamp_mode=clamp_mode, noise=nz)
        if hierarchical:
            with torch.no_grad():
                _f, _d, w = integrate(out, z_vals, noise)
                w = w.reshape(n * rays, steps) + 1e-5
                z = z_vals.reshape(n * rays, steps)
                z_fine = vr.sample_pdf(0.5 * (z[:, :-1] + z[:, 1:]), w[:, 1:-1], steps, det=False, u=importance_u)
                z_fine = z_fine.reshape(n, rays, steps, 1)
                pts_fine = rays_o.unsqueeze(2) + rays_d.unsqueeze(2) * z_fine
            out_fine = self.sample_voxel(img_v, seg_v, pts_fine.reshape(n, -1, 3)).reshape(n, rays, steps, -1)
            z_vals = torch.cat([z_fine, z_vals], 2)
            z_vals, order = torch.sort(z_vals, dim=2)
            out = torch.gather(torch.cat([out_fine, out], 2), 2, order.expand(-1, -1, -1, out.shape[-1]))
            noise = sigma_noise_fine.unsqueeze(-1) if sigma_noise_fine is not None else None
        feat, depth, weights = integrate(out, z_vals, noise)
        feat = feat.permute(0, 2, 1).reshape(n, -1, size, size)
        depth = depth.permute(0, 2, 1).reshape(n, 1, size, size)
        wsum = weights.sum(2).permute(0, 2, 1).reshape(n, 1, size, size)
        return feat, depth, wsum


def _as_channels_last(t):
    return t if t.stride(1) == 1 else t.contiguous(memory_format=torch.channels_last)


@persistence.persistent_class
class TriplaneSynthesisNetwork(torch.nn.Module):
    """`G.synthesis`."""

    def __init__(self, spec: GeneratorSpec):
        super().__init__()
        self.spec = spec
        self.w_dim = spec.w_dim
        self.img_resolution = spec.img_resolution
        self.img_channels = spec.img_channels
        self.seg_channels = spec.seg_channels
        self.render_size = spec.render_size
        self.voxel_block_resolutions = spec.voxel_resolutions()
        self.block_resolutions = spec.sr_resolutions()
        plane_ch = 3 * spec.plane_channels
        layer_kwargs = dict(layer_name='training.networks.SynthesisLayer')
        # fp16 blocks like the reference's SynthesisNetwork (inversion/networks.py:1168-1179: `use_fp16 = res >= fp16_resolution`,
        # conv_clamp 256 with fp16): the top `num_fp16_res` resolutions of the voxel chain and, when any, the super-resolution blocks
        conv_clamp = spec.conv_clamp if (spec.conv_clamp is not None or spec.num_fp16_res == 0) else 256
        fp16_from = (spec.plane_resolution >> max(spec.num_fp16_res - 1, 0)) if spec.num_fp16_res > 0 else None

        self.num_ws = 0
        for res in self.voxel_block_resolutions:
            cin = spec.voxel_width(res // 2) if res > 4 else 0
            block = VoxelBlock(cin, spec.voxel_width(res), w_dim=spec.w_dim, resolution=res, img_channels=plane_ch,
                               seg_channels=plane_ch, is_last=False, architecture='skip', conv_clamp=conv_clamp,
                               use_fp16=(fp16_from is not None and res >= max(fp16_from, 8)), **layer_kwargs)
            self.num_ws += block.num_conv
            setattr(self, f'vb{res}', block)

        # the skip images of the last voxel block are the tri-planes: have them written channels-last (the gather layout)
        getattr(self, f'vb{self.voxel_block_resolutions[-1]}').skip_channels_last = True
        self.renderer = TriplaneRenderer(spec)
        self.style_prefetch = True        # GPU inference: style kernels on a side stream (networks.prefetch_styles)

        widths = spec.sr_widths()
        cin = spec.feature_channels
        for res in self.block_resolutions:
            is_last = (res == self.img_resolution)
            block = networks.SegSynthesisBlock(cin, widths[res], w_dim=spec.w_dim, resolution=res,
                                               img_channels=spec.img_channels, seg_channels=spec.seg_channels,
                                               is_last=is_last, architecture='skip', conv_clamp=conv_clamp,
                                               use_fp16=(spec.num_fp16_res > 0), **layer_kwargs)
            self.num_ws += block.num_conv
            if is_last:
                self.num_ws += block.num_torgb
            setattr(self, f'b{res}', block)
            cin = widths[res]

    # -- pieces (public so that drivers / caches can call them separately) ----------------------------------
    def split_ws(self, ws):
        """Per-block w slices, StyleGAN2 convention (extract_shapes.py:113-124)."""
        with torch.autograd.profiler.record_function('split_ws'):
            misc.assert_shape(ws, [None, self.num_ws, self.w_dim])
            ws = ws.to(torch.float32)
            voxel_ws, block_ws, w_idx = [], [], 0
            for res in self.voxel_block_resolutions:
                block = getattr(self, f'vb{res}')
                voxel_ws.append(ws.narrow(1, w_idx, block.num_conv + block.num_torgb))
                w_idx += block.num_conv
            for res in self.block_resolutions:
                block = getattr(self, f'b{res}')
                block_ws.append(ws.narrow(1, w_idx, block.num_conv + block.num_torgb))
                w_idx += block.num_conv
        return voxel_ws, block_ws

    def backbone(self, voxel_ws, **block_kwargs):
        """ws -> (texture tri-plane, semantic tri-plane); pose independent."""
        x_v = img_v = seg_v = None
        blocks = [getattr(self, f'vb{res}') for res in self.voxel_block_resolutions]
        start, resume = 0, False
        # the 4^2 .. 16^2 / 32^2 blocks in one launch when that applies (GPU inference, default arithmetic, no hooks on them): csrc/lowres.hip
        grp = networks.lowres_group_forward(blocks, voxel_ws, **block_kwargs) if (voxel_ws and torch.is_tensor(voxel_ws[0]) and voxel_ws[0].is_cuda) else None
        if grp is not None:
            x_v, img_v, seg_v, start, resume = grp
        for i, (block, cur_ws) in enumerate(zip(blocks, voxel_ws)):
            if i < start:
                continue
            extra = dict(_resume_after_conv0=True) if (resume and i == start) else {}
            x_v, img_v, seg_v = block(x_v, img_v, cur_ws, condition_img=seg_v, **block_kwargs, **extra)
        return img_v, seg_v

    @contextlib.contextmanager
    def _pass_scope(self, ws, voxel_and_sr):
        """What surrounds the launches of one pass: the amax arena of the f16x3 arithmetic and — only while a hipGraph is being
        captured — the style / demodulation / head-folding launches of the given blocks on a side stream (they depend only on ws).
        With eager launches the host is the bottleneck and 43 extra events cost more than the overlap returns (measured: -11 % on
        the eager video driver, +2.4 % on the graphed renderer)."""
        voxel, sr = voxel_and_sr
        prefetch = (getattr(self, 'style_prefetch', True) and ws.is_cuda and not torch.is_grad_enabled()
                    and torch.cuda.is_current_stream_capturing() and not os.environ.get('IDE3D_NO_STYLE_PREFETCH'))
        side = None
        arena = networks.amax_arena(ws.shape[0], ws.device)
        arena.__enter__()
        try:
            if prefetch:
                side = networks.side_stream(ws.device)
                todo = [(getattr(self, f'vb{r}'), w) for r, w in voxel] + [(getattr(self, f'b{r}'), w) for r, w in sr]
                networks.prefetch_styles(todo, side)
            yield
        finally:
            arena.__exit__(None, None, None)
            if side is not None:
                networks.finish_prefetch(side)      # joins the side stream and drops the table even when a layer raised

    def planes(self, ws, noise_mode='const', force_fp32=False):
        """ws [N, num_ws, w_dim] -> (texture tri-plane, semantic tri-plane): `split_ws` + `backbone` as one call, so that drivers which
        cache the pose-independent tri-planes per seed (training/distributed_render.py, training/video_render.py) get the captured-graph
        rate for it too (training/graph_cache.py; fresh tensors per call)."""
        def impl(ws_, _c, _jitter, _planes):
            voxel_ws, _ = self.split_ws(ws_)
            with self._pass_scope(ws_, (list(zip(self.voxel_block_resolutions, voxel_ws)), [])):
                return self.backbone(voxel_ws, noise_mode=noise_mode, force_fp32=force_fp32)

        if torch.is_tensor(ws) and ws.is_cuda:
            return graph_cache.run(self, impl, ws, None, {}, noise_mode, (), force_fp32, False, None, {}, kind='backbone')
        return impl(ws, None, False, None)

    def superres(self, feat, block_ws, **block_kwargs):
        """[N, feat+seg, R, R] composited features -> (img, seg) at full resolution."""
        fc = self.spec.feature_channels
        size = self.block_resolutions[0] // 2
        if (size == 2 * feat.shape[-1] and size == 2 * feat.shape[-2] and feat.shape[-1] % 2 == 0 and networks._inference_on_gpu(feat)
                and networks._resample_init()):
            # one launch (csrc/resample.hip) instead of three ATen bilinear launches: same source-index rule and weights
            # (raw RGB and semantic logits as the two channel ranges of one tensor: the first block's skip up-sampler then takes both in one launch,
            # like every later block's, whose low-resolution pair is the previous dual head's output)
            x, img, seg = networks._resample_plugin.bilinear_up2_split(
                feat, [(0, fc), (0, self.img_channels), (fc, feat.shape[1] - fc)], adjacent=(1, 2))
        else:
            up = lambda t: torch.nn.functional.interpolate(t, size=(size, size), mode='bilinear', align_corners=False)
            x = up(feat[:, :fc])
            img = up(feat[:, :self.img_channels])
            seg = up(feat[:, fc:])
        for res, cur_ws in zip(self.block_resolutions, block_ws):
            x, img, seg = getattr(self, f'b{res}')(x, img, seg, cur_ws, **block_kwargs)
        return img, seg

    def forward(self, ws, c=None, render_params=None, noise_mode='const', return_seg=False, return_raw=False,
                return_dict=False, force_fp32=False, cond_img=None, ray_jitter=None, cached_planes=None, **unused):
        """ws [N, num_ws, w_dim], c [N, 25] = flattened cam2world (16) + intrinsics (9).

        On the GPU in inference a repeated call signature is captured into a hipGraph and replayed (training/graph_cache.py: the
        reference's per-image driver loops are bound by the host's ~80 launches per pass otherwise); outputs are fresh tensors
        either way.  `self.auto_graph = False`, `IDE3D_AUTO_GRAPH=0` or a forward hook anywhere in the tree keep every call eager."""
        assert c is not None, 'synthesis needs the 25-D camera label c'
        render_params = dict(render_params or {})
        flags = (bool(return_seg), bool(return_raw), bool(return_dict))

        def impl(ws_, c_, jitter_, planes_):
            return self._forward_impl(ws_, c_, render_params, noise_mode, flags, force_fp32, jitter_, planes_)

        if torch.is_tensor(ws) and ws.is_cuda:
            return graph_cache.run(self, impl, ws, c, render_params, noise_mode, flags, force_fp32, ray_jitter, cached_planes, unused)
        return impl(ws, c, ray_jitter, cached_planes)

    def _forward_impl(self, ws, c, render_params, noise_mode, flags, force_fp32, ray_jitter, cached_planes):
        """The pass itself, launch by launch (what `forward` runs eagerly, and what a capture records)."""
        return_seg, return_raw, return_dict = flags
        voxel_ws, block_ws = self.split_ws(ws)
        # `force_fp32` (viz/renderer.py:439 passes it) reaches the blocks; without fp16 blocks in the spec it changes nothing
        block_kwargs = dict(noise_mode=noise_mode, force_fp32=force_fp32)
        with self._pass_scope(ws, ([] if cached_planes is not None else list(zip(self.voxel_block_resolutions, voxel_ws)), list(zip(self.block_resolutions, block_ws)))):
            if cached_planes is not None:
                img_v, seg_v = cached_planes
            else:
                img_v, seg_v = self.backbone(voxel_ws, **block_kwargs)
            cam2world = c[:, :16].reshape(-1, 4, 4).to(torch.float32)
            feat, depth, wsum = self.renderer(
                img_v, seg_v, cam2world, fov=render_params.get('fov'), num_steps=render_params.get('num_steps'),
                ray_start=render_params.get('ray_start'), ray_end=render_params.get('ray_end'),
                nerf_noise=render_params.get('nerf_noise', 0.0), jitter=ray_jitter,
                hierarchical=render_params.get('hierarchical'), importance_u=render_params.get('importance_u'))
            img, seg = self.superres(feat, block_ws, **block_kwargs)
        img_raw = feat[:, :self.img_channels]
        if return_dict:
            return dict(image=img, image_seg=seg, image_raw=img_raw, image_depth=depth, planes=(img_v, seg_v))
        if return_seg and return_raw:
            return img, seg, img_raw
        if return_seg:
            return img, seg
        if return_raw:
            return img, img_raw
        return img


@persistence.persistent_class
class TriPlaneGenerator(torch.nn.Module):
    """`G`: mapping + synthesis, with the attributes the reference drivers read."""

    def __init__(self, spec: Optional[GeneratorSpec] = None, **spec_overrides):
        super().__init__()
        if spec is None:
            spec = GeneratorSpec(**spec_overrides)
        elif spec_overrides:
            spec = dataclasses.replace(spec, **spec_overrides)
        self.spec = spec
        self.z_dim, self.c_dim, self.w_dim = spec.z_dim, spec.c_dim, spec.w_dim
        self.img_resolution, self.img_channels = spec.img_resolution, spec.img_channels
        self.synthesis = TriplaneSynthesisNetwork(spec)
        self.num_ws = self.synthesis.num_ws
        self.mapping = networks.MappingNetwork(z_dim=spec.z_dim, c_dim=spec.c_dim, w_dim=spec.w_dim, num_ws=self.num_ws,
                                               num_layers=spec.mapping_layers)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return self.synthesis(ws, c=c, **synthesis_kwargs)


# ---- camera labels used by the reference drivers ----------------------------------------------------------

INTRINSICS = (4.2647, 0.0, 0.5, 0.0, 4.2647, 0.5, 0.0, 0.0, 1.0)     # gen_images.py:87,107


def conditioning_label(device='cpu'):
    """Frontal 25-D label the mapping network is conditioned on (gen_images.py:87)."""
    pose = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 2.7, 0, 0, 0, 1]
    return vr.device_const(pose + list(INTRINSICS), torch.float32, device).reshape(1, -1).clone()


def camera_label(yaw, pitch=math.pi * 0.5, radius=2.7, device='cpu'):
    """25-D label of the gen_images.py pose loop (:104-107): camera on a sphere looking at the origin."""
    cam, _phi, _theta = vr.sample_camera_positions(device, n=1, r=radius, horizontal_mean=yaw + math.pi * 0.5,
                                                   vertical_mean=pitch, mode=None)
    c2w = vr.create_cam2world_matrix(-cam, cam, device=device).reshape(1, -1)
    return torch.cat((c2w, vr.device_const(INTRINSICS, torch.float32, device).reshape(1, -1)), -1)


class GraphedRenderer:
    """hipGraph replay of `G.mapping` + `G.synthesis` for a fixed batch size (MI355X: ~400 short launches per
    batch; replaying one captured graph removes the per-launch host cost).  Inputs are copied into static buffers,
    outputs are views of static buffers that the next call overwrites.

        run = GraphedRenderer(G, batch=4, device=dev)          # warms up, then captures
        img, seg = run(z, c_cond, c_cam)                       # z [B, z_dim] float64/32, labels [B, 25]
        img, seg = run(z, c_cond, c_cam, jitter=u)             # u [B, R*R, S]: the stratified-jitter draws of this call

    Stratified jitter (the reference draws it on every call, volumetric_rendering.py:113): the captured pass reads a static
    buffer `self.jitter`; a call refills it with fresh U[0,1) draws on the device, or with the caller's draws (parity
    tests, `bench.py`'s parity check).  `ray_jitter=False` captures the pass without jitter.

    `conv_arithmetic=` ('fp32' | 'bf16x6' | 'f16x3' | 'bf16x3'): the arithmetic of the 3x3 convolutions INSIDE this graph (the kernels are
    chosen while it is captured; the process-wide setting is restored afterwards).  None = whatever `hip_plugin.conv_arithmetic()` is in
    force.  See that function's docstring for when a split arithmetic may be selected.

    `static_labels=True`: a label tensor (c_cond / c_cam) that is the same object at the same `_version` as in the previous call is not
    copied again.  Opt-in, because `_version` does not see every write (`.data`, numpy views of a CPU tensor, custom kernels): with the
    default (False) every call copies its labels.

    `z` may be a pinned host tensor; it is copied asynchronously into the static buffer.  The caller must not refill that host buffer
    before the copy has run: `self.z_copied` is an event recorded right after it (`run.z_copied.synchronize()`), or pass a fresh tensor
    per call as `bench.py` does.
    """

    def __init__(self, G, batch, device, truncation_psi=1.0, noise_mode='const', warmup=3, ray_jitter=None, conv_arithmetic=None,
                 static_labels=False):
        self.G = G
        self.psi, self.noise_mode = truncation_psi, noise_mode
        self.static_labels = bool(static_labels)
        self.z_copied = torch.cuda.Event() if torch.device(device).type == 'cuda' else None
        restore = None
        if conv_arithmetic is not None:
            from torch_utils import hip_plugin
            restore = hip_plugin.conv_arithmetic()
            hip_plugin.conv_arithmetic(conv_arithmetic)
        try:
            self._capture(G, batch, device, warmup, ray_jitter)
        finally:
            if restore is not None:
                hip_plugin.conv_arithmetic(restore)

    def _capture(self, G, batch, device, warmup, ray_jitter):
        self.z = torch.zeros([batch, G.z_dim], dtype=torch.float32, device=device)
        self.c_cond = conditioning_label(device).repeat(batch, 1)
        self.c_cam = conditioning_label(device).repeat(batch, 1)
        sp = G.synthesis.spec
        if ray_jitter is False:
            self.jitter = None
        else:
            self.jitter = torch.rand([batch, sp.render_size ** 2, sp.num_steps], device=device)
            if ray_jitter is not None:
                self.jitter.copy_(ray_jitter)
        self._fixed_jitter = ray_jitter is not None and ray_jitter is not False
        self.graph = None
        from torch_utils import hip_plugin
        stream = hip_plugin.private_stream(device, 'graph capture')          # never one of torch's 32 pooled handles (hip_plugin.private_stream)
        stream.wait_stream(torch.cuda.current_stream(device))
        # Every launch of the warm-up and of the capture takes its scratch memory (packed weights, split-K partials, the mapping
        # kernel's barrier counter) from workspaces owned by THIS object (hip_plugin.workspace_scope), not by a stream handle that
        # another graph or an eager caller may be handed as well; the replays reuse the copies packed during warm-up.
        scope = self._scope(device)
        # the lock covers the warm-up too: all captures of a device share the 'graph capture' stream, and warm-up launches on it
        # while another thread holds it in capture mode would be recorded into (or invalidate) that thread's graph
        with hip_plugin.capture_lock:
            with torch.cuda.stream(stream), torch.no_grad(), scope, graph_cache.disabled():      # this object IS the capture: no automatic one inside it
                for _ in range(warmup):
                    self._body()
            torch.cuda.current_stream(device).wait_stream(stream)
            torch.cuda.synchronize(device)
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), scope, torch.cuda.graph(graph, stream=stream):
                self.out = self._body()
        self.graph = graph
        self._stream = stream

    def _scope(self, device):
        import contextlib
        if torch.device(device).type != 'cuda':
            return contextlib.nullcontext()
        from torch_utils import hip_plugin
        return hip_plugin.workspace_scope(self)

    def _body(self):
        ws = self.G.mapping(self.z, self.c_cond, truncation_psi=self.psi)
        return self.G.synthesis(ws, c=self.c_cam, noise_mode=self.noise_mode, return_seg=True,
                                ray_jitter=(False if self.jitter is None else self.jitter))

    def _unchanged(self, name, t):
        """True when `t` is the very tensor object, at the same version, that was copied into the static buffer `name` last time (the
        object is kept referenced: a freed tensor's id and address can come back with different contents)."""
        if not self.static_labels:
            return False
        last = getattr(self, '_last_' + name, None)
        if last is not None and last[0] is t and last[1] == t._version:
            return True
        setattr(self, '_last_' + name, (t, t._version))
        return False

    def __call__(self, z, c_cond=None, c_cam=None, jitter=None):
        # every launch in front of the replay is a few microseconds of GPU timeline: z may be a (pinned) host tensor — one copy straight
        # into the static buffer, dtype conversion included — and (static_labels=True) labels that are the tensors of the previous call are not copied again
        self.z.copy_(z, non_blocking=True)
        if self.z_copied is not None:
            self.z_copied.record()
        if c_cond is not None and not self._unchanged('c_cond', c_cond):
            self.c_cond.copy_(c_cond)
        if c_cam is not None and not self._unchanged('c_cam', c_cam):
            self.c_cam.copy_(c_cam)
        if self.jitter is not None:
            if jitter is not None:
                self.jitter.copy_(jitter)
            elif not self._fixed_jitter:
                self.jitter.uniform_()
        self.graph.replay()
        return self.out
