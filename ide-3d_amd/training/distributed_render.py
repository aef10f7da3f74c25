"""Batched multi-seed / multi-camera rendering sharded over the GPUs of one node.

The reference renders strictly one image per `G.synthesis` call on one GPU (gen_images.py:88-114,
gen_videos.py:114-139).  Every (seed, camera) pair is independent, so the MI355X build shards the work list
one rank per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI) with NO collective on the data path;
the only exchange is the final gather of finished uint8 frames to rank 0 (config 4 of BASELINE.json: 64 seeds x 8
poses -> grid video).  Frames travel as uint8 RGB | coloured-segmentation (6 bytes / pixel) instead of fp32 image +
19-channel logits (88 bytes / pixel): the argmax + palette look-up of `mask2color` (dnnlib/seg_tools.py:75-81) and the
uint8 conversion of `layout_grid` (dnnlib/util.py:637) run on the producing GPU (`csrc/frame.hip`).

Work assignment keeps all poses of a seed on one rank (the mapping network and — with `cache_backbone` — the
pose-independent tri-planes are evaluated once per seed).
"""

import math
from typing import List, Sequence, Tuple

import numpy as np
import torch

from torch_utils import custom_ops
from training import triplane

# 19-class face-parsing palette (dnnlib/seg_tools.py:13-32)
PALETTE = ((0, 0, 0), (204, 0, 0), (76, 153, 0), (204, 204, 0), (51, 51, 255), (204, 0, 204), (0, 255, 255), (255, 204, 204),
           (102, 51, 0), (255, 0, 0), (102, 204, 0), (255, 255, 0), (0, 0, 153), (0, 0, 204), (255, 51, 153), (0, 204, 204),
           (0, 51, 0), (255, 153, 51), (0, 204, 0))

_frame_plugin = None


def palette_tensor(num_classes, device):
    pal = [PALETTE[i % len(PALETTE)] for i in range(num_classes)]
    return torch.tensor(pal, dtype=torch.uint8, device=device)


def frames_u8(img, seg, palette=None, out=None):
    """[N, 3, H, W] image in [-1, 1] + [N, K, H, W] logits -> uint8 [N, H, 2W, 3] (RGB | palette[argmax seg]).
    `out`: write into this (contiguous, same shape) tensor instead of a new one — the send buffers of `OverlappedFrameGather`."""
    global _frame_plugin
    if palette is None:
        palette = palette_tensor(seg.shape[1], img.device)
    if img.device.type == 'cuda' and img.dtype == torch.float32 and img.shape[-1] % 4 == 0:
        if _frame_plugin is None:
            _frame_plugin = custom_ops.get_plugin(module_name='frame_plugin', sources=['frame.hip'])
        return _frame_plugin.frame_u8(img, seg.float(), palette, out=out)
    rgb = (img.float() * 127.5 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
    col = palette[torch.argmax(seg, dim=1)]
    if out is not None:
        return torch.cat([rgb, col], dim=2, out=out)
    return torch.cat([rgb, col], dim=2).contiguous()


class OverlappedFrameGather:
    """Double-buffered, asynchronous gather of fixed-size uint8 frame batches to rank `dst` ("gather of batch k
    overlaps render of k+1").  Step k writes its frames into `slot()` — a send buffer whose previous gather (step k - depth) has been
    waited for — and `submit()`s it: `dist.gather(..., async_op=True)` runs on the communication stream (RCCL's own stream; a gloo
    worker thread on CPU), so the compute stream goes straight on to step k + 1.  Only the wait in `slot()` / `drain()` orders the two
    streams again.  On `dst`, `received(i)` is the list of per-rank buffers of submission i (valid after `drain()` or after the
    matching `wait(i)`); a buffer is reused `depth` submissions later.  Nothing like it upstream: the reference renders on one GPU."""

    def __init__(self, shape, device, rank, world, dst=0, depth=2, dtype=torch.uint8, checksums=0):
        """`checksums=K` (> 0): a position-weighted 64-bit checksum of every buffer sent (all ranks) and received (on `dst`) is kept for the
        last K submissions — `sent_checksums()` / `received_checksums()` — so that a whole run can be verified, not a sample of it
        (`bench.py gather_check`).  The sums of a received buffer are formed on a side stream once its gather has completed; the compute
        stream only waits for them before it reuses that buffer `depth` submissions later."""
        import torch.distributed as dist
        self._dist, self.rank, self.world, self.dst, self.depth = dist, rank, world, dst, depth
        self.send = [torch.empty(shape, dtype=dtype, device=device) for _ in range(depth)]
        self.recv = [[torch.empty(shape, dtype=dtype, device=device) for _ in range(world)] if rank == dst else None for _ in range(depth)]
        self.work = [None] * depth
        self.submitted = 0
        self.cap = int(checksums)
        if self.cap:
            nbytes = self.send[0].numel() * self.send[0].element_size()
            assert nbytes % 8 == 0, 'checksums view the buffers as 64-bit words'
            self._mult = torch.arange(1, nbytes // 8 + 1, dtype=torch.int64, device=device) * 2 + 1          # odd multipliers: position matters
            self._sent = torch.zeros([self.cap], dtype=torch.int64, device=device)
            self._rcvd = torch.zeros([self.cap, world], dtype=torch.int64, device=device) if rank == dst else None
            self._pending = [None] * depth          # submission number whose received buffers have not been summed yet
            cuda = torch.device(device).type == 'cuda'
            self._check_stream = torch.cuda.Stream(device=device) if cuda else None
            self._checked = [None] * depth          # event: sums of the buffers in slot b are done (the buffer may be overwritten)

    def _sum64(self, buf):
        return (buf.reshape(-1).view(torch.int64) * self._mult).sum()

    def _wait_slot(self, b):
        """Order the current stream behind the gather of slot b (if one is in flight).  With checksums: the sums of what arrived are formed
        on the side stream, which waits for the COLLECTIVE only — not for whatever the compute stream has queued (the render of the current
        step) — so they run beside the rendering."""
        w = self.work[b]
        if w is None:
            return
        self.work[b] = None
        i = self._pending[b] if self.cap else None
        if i is not None:
            self._pending[b] = None
            if self._rcvd is not None:
                if self._check_stream is not None:
                    with torch.cuda.stream(self._check_stream):
                        w.wait()
                        for r in range(self.world):
                            self._rcvd[i % self.cap, r] = self._sum64(self.recv[b][r])
                        ev = torch.cuda.Event(); ev.record(self._check_stream)
                    self._checked[b] = ev
                    w.wait()
                    return
                w.wait()
                for r in range(self.world):
                    self._rcvd[i % self.cap, r] = self._sum64(self.recv[b][r])
                return
        w.wait()

    def wait(self, i):
        """Wait (stream-ordered on the GPU) for submission number i, if it is still in flight."""
        b = i % self.depth
        if self.submitted - i <= self.depth:
            self._wait_slot(b)

    def slot(self):
        """The send buffer of the next submission; its previous use has completed when this returns."""
        b = self.submitted % self.depth
        self._wait_slot(b)
        if self.cap and self._checked[b] is not None:
            torch.cuda.current_stream(self.send[0].device).wait_event(self._checked[b])       # the receive buffers of this slot are about to be overwritten
            self._checked[b] = None
        return self.send[b]

    def submit(self):
        """Start the gather of the buffer `slot()` returned; returns the submission number."""
        b = self.submitted % self.depth
        if self.cap:
            self._sent[self.submitted % self.cap] = self._sum64(self.send[b])
            self._pending[b] = self.submitted
        self.work[b] = self._dist.gather(self.send[b], self.recv[b], dst=self.dst, async_op=True)
        self.submitted += 1
        return self.submitted - 1

    def received(self, i):
        return self.recv[i % self.depth]

    def drain(self):
        for b in range(self.depth):
            self._wait_slot(b)
        if self.cap and self._check_stream is not None:
            torch.cuda.current_stream(self.send[0].device).wait_stream(self._check_stream)

    def sent_checksums(self, first, count):
        """int64 [count]: checksums of this rank's submissions first .. first + count - 1 (the last `checksums` submissions are kept)."""
        assert self.cap and count <= self.cap and first + count <= self.submitted and first >= self.submitted - self.cap
        return self._sent[torch.arange(first, first + count, device=self._sent.device) % self.cap]

    def received_checksums(self, first, count):
        """int64 [count, world] on `dst` (after `drain()`): checksums of what arrived from every rank in those submissions."""
        assert self.cap and self._rcvd is not None and count <= self.cap and first + count <= self.submitted and first >= self.submitted - self.cap
        return self._rcvd[torch.arange(first, first + count, device=self._rcvd.device) % self.cap]


def shard_items(num_seeds: int, num_poses: int, rank: int, world: int) -> List[Tuple[int, int]]:
    """(seed index, pose index) pairs of this rank: seeds are dealt round-robin, every pose of a seed stays together."""
    return [(s, p) for s in range(rank, num_seeds, world) for p in range(num_poses)]


def item_index(seed_idx: int, pose_idx: int, num_poses: int) -> int:
    return seed_idx * num_poses + pose_idx


@torch.no_grad()
def render_items(G, seeds: Sequence[int], yaws: Sequence[float], items: Sequence[Tuple[int, int]], device, batch: int = 4,
                 truncation_psi: float = 1.0, noise_mode: str = 'const', cache_backbone: bool = True, jitter_seed=None):
    """Render the given (seed, pose) pairs; returns uint8 frames [len(items), H, 2W, 3] in item order."""
    if len(items) == 0:
        res = G.img_resolution
        return torch.empty([0, res, 2 * res, 3], dtype=torch.uint8, device=device)
    cond = triplane.conditioning_label(device)
    cams = {p: triplane.camera_label(yaws[p], device=device) for p in sorted({p for _, p in items})}
    palette = palette_tensor(G.synthesis.seg_channels, device)
    ws_cache, plane_cache, out = {}, {}, []
    plane_buf, buf_seed = None, None
    for start in range(0, len(items), batch):
        chunk = items[start:start + batch]
        for s, _ in chunk:
            if s not in ws_cache:
                z = torch.from_numpy(np.random.RandomState(seeds[s]).randn(1, G.z_dim)).to(device)
                ws_cache[s] = G.mapping(z, cond, truncation_psi=truncation_psi)
                if cache_backbone:
                    plane_cache[s] = G.synthesis.planes(ws_cache[s], noise_mode=noise_mode)
        ws = torch.cat([ws_cache[s] for s, _ in chunk])
        c = torch.cat([cams[p] for _, p in chunk])
        planes = None
        if cache_backbone:
            # the batch's tri-planes live in ONE pair of buffers for the whole job: a row is rewritten only when its seed changes (all
            # poses of a seed are consecutive items), and `G.synthesis` sees the same addresses in every call — it reads cached
            # tri-planes in place, so its captured pass (training/graph_cache.py) is keyed on them
            if plane_buf is None:
                like = plane_cache[chunk[0][0]]
                plane_buf = tuple(torch.empty([batch, *t.shape[1:]], dtype=t.dtype, device=t.device).contiguous(memory_format=torch.channels_last)
                                  for t in like)
                buf_seed = [None] * batch
            for j, (s, _) in enumerate(chunk):
                if buf_seed[j] != s:
                    for k in range(2):
                        plane_buf[k][j].copy_(plane_cache[s][k][0])
                    buf_seed[j] = s
            planes = (plane_buf[0][:len(chunk)], plane_buf[1][:len(chunk)])
        jit = None
        if jitter_seed is not None:      # reproducible stratified jitter (same draws whatever the sharding)
            g = G.synthesis
            jit = torch.stack([torch.rand([g.render_size ** 2, g.spec.num_steps],
                                          generator=torch.Generator().manual_seed(jitter_seed + item_index(s, p, len(yaws)))) for s, p in chunk]).to(device)
        img, seg = G.synthesis(ws, c=c, noise_mode=noise_mode, return_seg=True, cached_planes=planes, ray_jitter=jit)
        out.append(frames_u8(img, seg, palette))
        # drop cached tri-planes of seeds that are finished
        done = {s for s, _ in items[:start + batch]} - {s for s, _ in items[start + batch:]}
        for s in done:
            plane_cache.pop(s, None)
    return torch.cat(out)


def gather_frames(frames: torch.Tensor, counts: Sequence[int], rank: int, world: int, dst: int = 0):
    """Gather per-rank uint8 frame stacks (different lengths allowed) on `dst`; returns the list of per-rank tensors
    on `dst`, None elsewhere.  One direct peer->root transfer per rank (xGMI is fully connected)."""
    import torch.distributed as dist
    if world == 1:
        return [frames]
    cap = max(counts)
    pad = frames
    if frames.shape[0] < cap:
        pad = torch.cat([frames, frames.new_zeros([cap - frames.shape[0], *frames.shape[1:]])])
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return [b[:counts[r]] for r, b in enumerate(bufs)]


@torch.no_grad()
def render_grid_sharded(G, seeds: Sequence[int], yaws: Sequence[float], device, rank: int = 0, world: int = 1, batch: int = 4,
                        **render_kwargs):
    """Config 4: every rank renders its shard, rank 0 receives all frames ordered as [seed][pose].
    Returns uint8 [len(seeds) * len(yaws), H, 2W, 3] on rank 0, None on other ranks."""
    items = shard_items(len(seeds), len(yaws), rank, world)
    frames = render_items(G, seeds, yaws, items, device, batch=batch, **render_kwargs)
    counts = [len(shard_items(len(seeds), len(yaws), r, world)) for r in range(world)]
    parts = gather_frames(frames, counts, rank, world)
    if parts is None:
        return None
    res = frames.new_empty([len(seeds) * len(yaws), *frames.shape[1:]])
    for r, part in enumerate(parts):
        idx = [item_index(s, p, len(yaws)) for s, p in shard_items(len(seeds), len(yaws), r, world)]
        if idx:
            res[torch.tensor(idx, device=res.device)] = part.to(res.device)
    return res
