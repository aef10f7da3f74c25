"""Grid video sweep: the frame loop of `gen_videos.gen_interp_video` (gen_videos.py:66-131) on the HIP rendering path.

Same schedule as the reference — keyframe latents per grid cell, `scipy.interpolate.interp1d` of the keyframe `ws` over
time (`kind='cubic'`, `wraps` periodic copies), one look-at camera per frame sweeping yaw by sin and pitch by cos
(gen_videos.py:118-122), `image_mode='image_seg'` frames = RGB | colourised segmentation laid out as a grid — but
  * all grid cells of a frame are rendered in ONE batched `G.synthesis` call (the reference renders them one by one),
  * when a cell's `ws` does not change over time (one keyframe per cell, the BASELINE config-3 case) its two tri-planes
    are computed once and reused for every pose (`cached_planes`: the `vb*` backbone is 2/3 of the frame's FLOPs),
  * colour mapping + grid layout happen on the GPU in uint8 (`frames_u8` / `ide3d_frame_u8`) — the encoder
    (imageio / libx264, gen_videos.py:108,131) is a host-side consumer and out of scope: frames are yielded.
"""

import math
import threading
import weakref
from typing import Iterable, Iterator, Sequence, Tuple

import numpy as np
import scipy.interpolate
import torch

from training import distributed_render as dr
from training import triplane
from training.volumetric_rendering import LookAtPoseSampler, device_const

INTRINSICS = ((4.2647, 0, 0.5), (0, 4.2647, 0.5), (0, 0, 1))          # gen_videos.py:90


def sweep_pose(frame_idx: int, total_frames: int, lookat, radius: float = 2.7, yaw_range: float = 0.5, pitch_range: float = 0.25,
               device='cpu') -> torch.Tensor:
    """25-D camera label of frame `frame_idx` (gen_videos.py:116-125)."""
    t = 2 * math.pi * frame_idx / total_frames
    cam2world = LookAtPoseSampler.sample(math.pi / 2 - yaw_range * np.sin(t), math.pi / 2 - 0.05 + pitch_range * np.cos(t),
                                         lookat, radius=radius, device=device)
    intr = device_const([v for row in INTRINSICS for v in row], torch.float32, device).reshape(3, 3)
    return torch.cat([cam2world.reshape(-1, 16), intr.reshape(-1, 9)], 1)


def interpolate_ws(ws_keyframes: np.ndarray, w_frames: int, kind: str = 'cubic', wraps: int = 2) -> np.ndarray:
    """ws of one grid cell for every frame: [num_keyframes * w_frames, num_ws, w_dim] (gen_videos.py:95-104,127-128)."""
    k = ws_keyframes.shape[0]
    x = np.arange(-k * wraps, k * (wraps + 1))
    y = np.tile(ws_keyframes, [wraps * 2 + 1, 1, 1])
    interp = scipy.interpolate.interp1d(x, y, kind=kind, axis=0)
    return interp(np.arange(k * w_frames) / w_frames).astype(np.float32)          # all frames in one evaluation (= the per-frame calls of gen_videos.py:127, bit for bit)


_plane_pool = weakref.WeakKeyDictionary()          # synthesis module -> [[buffers, in_use], ...]: pairs of buffers static tri-planes are kept in
_plane_pool_lock = threading.Lock()


def _checkout_plane_buffers(synthesis, planes):
    """The job's static tri-planes, copied into buffers that outlive the job: `G.synthesis` reads cached tri-planes in place and keys its
    captured pass on their addresses (training/graph_cache.py), so the next video job of this generator replays the pass the previous one
    captured instead of capturing its own (a capture costs ~3 passes + a device synchronisation: 10 % of a 120-frame job).
    A pair of buffers belongs to ONE live job at a time (two interleaved generators on one G shared a pair and the second
    overwrote the planes the first was still rendering from): a job checks a pair out and `_release_plane_buffers` hands it back when the
    generator finishes or is closed; a second live job of the same shape gets its own pair.  Returns (planes to render from, token)."""
    if not planes[0].is_cuda:
        return planes, None
    with _plane_pool_lock:
        pool = _plane_pool.setdefault(synthesis, [])
        ent = next((e for e in pool if not e[1] and all(b.shape == t.shape and b.stride() == t.stride() and b.device == t.device and b.dtype == t.dtype
                                                        for b, t in zip(e[0], planes))), None)
        if ent is None:
            ent = [tuple(torch.empty_like(t) for t in planes), False]          # empty_like keeps the channels-last strides
            pool.append(ent)
            del pool[:-8]                                                      # bound: idle pairs of shapes no job uses any more
        ent[1] = True
    for b, t in zip(ent[0], planes):
        b.copy_(t)
    return ent[0], ent


def _release_plane_buffers(token):
    if token is not None:
        with _plane_pool_lock:
            token[1] = False


def layout_u8(frames: torch.Tensor, grid_w: int, grid_h: int) -> torch.Tensor:
    """uint8 [grid_h * grid_w, H, W', 3] cell frames -> [grid_h * H, grid_w * W', 3] (layout_grid, gen_videos.py:24-38)."""
    n, h, w, c = frames.shape
    assert n == grid_w * grid_h
    return frames.reshape(grid_h, grid_w, h, w, c).permute(0, 2, 1, 3, 4).reshape(grid_h * h, grid_w * w, c)


@torch.no_grad()
def gen_interp_frames(G, seeds: Sequence[int], shuffle_seed=None, w_frames: int = 60 * 4, kind: str = 'cubic',
                      grid_dims: Tuple[int, int] = (1, 1), num_keyframes=None, wraps: int = 2, psi: float = 1, truncation_cutoff=14,
                      cfg: str = 'FFHQ', image_mode: str = 'image_seg', device=torch.device('cuda'), noise_mode: str = 'const',
                      cache_static_planes: bool = True, **synthesis_kwargs) -> Iterator[torch.Tensor]:
    """Yields one uint8 grid frame [grid_h * H, grid_w * W', 3] per video frame (W' = 2 W for 'image_seg', else W).
    `synthesis_kwargs` go to `G.synthesis` (e.g. `ray_jitter=False` for reproducible frames: like the reference, the
    default draws fresh stratified jitter for every frame)."""
    grid_w, grid_h = grid_dims
    cells = grid_w * grid_h
    if num_keyframes is None:
        if len(seeds) % cells != 0:
            raise ValueError('Number of input seeds must be divisible by grid W*H')
        num_keyframes = len(seeds) // cells
    if image_mode not in ('image', 'image_seg'):
        raise ValueError("image_mode must be 'image' or 'image_seg'")
    all_seeds = np.array([seeds[i % len(seeds)] for i in range(num_keyframes * cells)], dtype=np.int64)
    if shuffle_seed is not None:
        np.random.RandomState(seed=shuffle_seed).shuffle(all_seeds)
    lookat = torch.tensor([0, 0, 0.2] if cfg == 'FFHQ' else [0, 0, 0], dtype=torch.float32, device=device)

    zs = torch.from_numpy(np.stack([np.random.RandomState(int(s)).randn(G.z_dim) for s in all_seeds])).to(device).float()
    front = LookAtPoseSampler.sample(math.pi / 2, math.pi / 2, lookat, radius=2.7, device=device)
    c_front = torch.cat([front.reshape(-1, 16), device_const([v for row in INTRINSICS for v in row], torch.float32, device).reshape(-1, 9)], 1)
    ws = G.mapping(zs, c_front.repeat(len(zs), 1), truncation_psi=psi, truncation_cutoff=truncation_cutoff)
    ws = ws.reshape(grid_h, grid_w, num_keyframes, *ws.shape[1:]).cpu().numpy()
    total = num_keyframes * w_frames
    # [cell][frame] ws, uploaded once
    ws_frames = torch.from_numpy(np.stack([interpolate_ws(ws[yi][xi], w_frames, kind, wraps)
                                           for yi in range(grid_h) for xi in range(grid_w)])).to(device)
    # one keyframe per cell: the spline through copies of one point is that point (up to interpolation round-off)
    static = cache_static_planes and num_keyframes == 1 and bool(torch.allclose(ws_frames, ws_frames[:, :1].expand_as(ws_frames), atol=1e-5))
    planes = token = None
    if static:
        planes, token = _checkout_plane_buffers(G.synthesis, G.synthesis.planes(ws_frames[:, 0], noise_mode=noise_mode))
    palette = dr.palette_tensor(G.synthesis.seg_channels, device)
    try:
        for frame_idx in range(total):
            c = sweep_pose(frame_idx, total, lookat, device=device).repeat(cells, 1)
            img, seg = G.synthesis(ws_frames[:, frame_idx], c=c, noise_mode=noise_mode, return_seg=True, cached_planes=planes, **synthesis_kwargs)
            if image_mode == 'image_seg':
                cell_frames = dr.frames_u8(img, seg, palette)
            else:
                cell_frames = (img.float() * 127.5 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
            yield layout_u8(cell_frames, grid_w, grid_h)
    finally:
        _release_plane_buffers(token)          # also on generator.close() / garbage collection of an abandoned iterator


# ---- dependency-free video sink --------------------------------------------------------------------------------------
# The reference encodes with imageio / libx264 (gen_videos.py:108,139), which is not installable here.  YUV4MPEG2 is the
# uncompressed stream every encoder reads (`ffmpeg -i out.y4m out.mp4`, `mpv out.y4m`), needs no library and keeps the
# colour conversion on the device that rendered the frame.

def rgb_u8_to_yuv444(frame: torch.Tensor) -> torch.Tensor:
    """uint8 [H, W, 3] RGB -> uint8 [3, H, W] planar Y'CbCr (BT.601, limited range, the integer matrix of ITU-T T.871 scaled to
    16..235 / 16..240): Y = 16 + (66 R + 129 G + 25 B + 128) >> 8, Cb = 128 + (-38 R - 74 G + 112 B + 128) >> 8,
    Cr = 128 + (112 R - 94 G - 18 B + 128) >> 8."""
    assert frame.dtype == torch.uint8 and frame.ndim == 3 and frame.shape[2] == 3
    rgb = frame.to(torch.int32)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    y = 16 + ((66 * r + 129 * g + 25 * b + 128) >> 8)
    cb = 128 + ((-38 * r - 74 * g + 112 * b + 128) >> 8)
    cr = 128 + ((112 * r - 94 * g - 18 * b + 128) >> 8)
    return torch.stack([y, cb, cr]).clamp_(0, 255).to(torch.uint8)


def write_y4m(frames: Iterable[torch.Tensor], path: str, fps: int = 60) -> int:
    """Write uint8 [H, W, 3] RGB frames (any device, e.g. what `gen_interp_frames` yields) as a YUV4MPEG2 4:4:4 stream.
    Returns the number of frames written."""
    count = 0
    with open(path, 'wb') as f:
        for frame in frames:
            if count == 0:
                h, w = int(frame.shape[0]), int(frame.shape[1])
                f.write(f'YUV4MPEG2 W{w} H{h} F{int(fps)}:1 Ip A1:1 C444\n'.encode('ascii'))
            assert tuple(frame.shape[:2]) == (h, w), 'all frames of a stream have one size'
            f.write(b'FRAME\n')
            f.write(rgb_u8_to_yuv444(frame).cpu().numpy().tobytes())
            count += 1
    return count


def read_y4m(path: str):
    """(header dict, uint8 array [frames, 3, H, W]) of a 4:4:4 stream written by `write_y4m` (tests, round trips)."""
    with open(path, 'rb') as f:
        head = f.readline().decode('ascii').split()
        assert head[0] == 'YUV4MPEG2' and 'C444' in head
        info = {t[0]: t[1:] for t in head[1:]}
        w, h = int(info['W']), int(info['H'])
        frames = []
        while True:
            line = f.readline()
            if not line:
                break
            assert line.startswith(b'FRAME')
            frames.append(np.frombuffer(f.read(3 * h * w), dtype=np.uint8).reshape(3, h, w))
    return info, np.stack(frames) if frames else np.zeros((0, 3, h, w), np.uint8)
