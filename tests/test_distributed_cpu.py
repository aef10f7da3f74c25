"""The N > 1 path on CPU: two `gloo` processes shard (seed, pose) pairs, gather uint8 frames on rank 0, and the result
must equal the single-process render byte for byte ("1/2/4/8-rank gather equals single-rank result")."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build_generator():
    from training import triplane
    torch.manual_seed(0)
    G = triplane.TriPlaneGenerator(triplane.tiny_spec()).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        for name, p in G.named_parameters():
            if name.endswith('noise_strength'):
                p.copy_(torch.randn([], generator=g) * 0.2)
    return G


def _worker(rank, world, port, seeds, yaws, out_path):
    for p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from training import distributed_render as dr
    torch.set_num_threads(2)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    G = _build_generator()
    frames = dr.render_grid_sharded(G, seeds, yaws, torch.device('cpu'), rank=rank, world=world, batch=2, jitter_seed=1234)
    if rank == 0:
        np.save(out_path, frames.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_items_partition():
    from training import distributed_render as dr
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            items = dr.shard_items(5, 3, r, world)
            assert all(s % world == r for s, _ in items)              # a seed never straddles ranks
            seen += [dr.item_index(s, p, 3) for s, p in items]
        assert sorted(seen) == list(range(15))


def test_two_rank_gather_equals_single_rank(tmp_path):
    from training import distributed_render as dr
    seeds, yaws = [3, 4, 5], [-0.4, 0.3]
    G = _build_generator()
    ref = dr.render_grid_sharded(G, seeds, yaws, torch.device('cpu'), batch=2, jitter_seed=1234).numpy()
    assert ref.shape == (6, 64, 128, 3) and ref.dtype == np.uint8
    # caching the pose-independent tri-planes changes only the batch composition of the backbone convolutions:
    # identical up to fp32 summation order, i.e. at most a few uint8 quantisation / argmax flips
    ref_nc = dr.render_grid_sharded(G, seeds, yaws, torch.device('cpu'), batch=3, jitter_seed=1234, cache_backbone=False).numpy()
    assert (ref != ref_nc).mean() < 5e-3
    out = str(tmp_path / 'frames.npy')
    mp.spawn(_worker, args=(2, _free_port(), seeds, yaws, out), nprocs=2, join=True)
    got = np.load(out)
    assert np.array_equal(got, ref)


def test_frames_u8_cpu_matches_oracle():
    from oracle import ops as oracle_ops
    from training import distributed_render as dr
    g = torch.Generator().manual_seed(0)
    img = torch.randn(2, 3, 8, 12, generator=g); seg = torch.randn(2, 19, 8, 12, generator=g)
    assert np.array_equal(dr.frames_u8(img, seg).numpy(), oracle_ops.frame_u8(img, seg))
    assert np.array_equal(np.array(dr.PALETTE, dtype=np.uint8), oracle_ops.PALETTE)


def _overlap_worker(rank, world, port, out_path):
    for p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from training import distributed_render as dr
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    og = dr.OverlappedFrameGather([2, 4, 8, 3], torch.device('cpu'), rank, world, depth=2)
    seen = []
    for step in range(5):                          # more submissions than buffers: every buffer is reused at least once
        og.slot().fill_(step * 10 + rank)          # slot() has waited for the buffer's previous gather
        i = og.submit()
        assert i == step
        if rank == 0 and step >= 1:                # read step - 1 while step is in flight
            og.wait(step - 1)
            seen.append([int(t.unique().item()) for t in og.received(step - 1)])
    og.drain()
    if rank == 0:
        seen.append([int(t.unique().item()) for t in og.received(4)])
        np.save(out_path, np.array(seen))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_frame_gather_order_and_contents(tmp_path):
    """The double-buffered asynchronous gather bench.py's N > 1 step uses: submission i delivers rank r's buffer of step i, in order,
    also when a buffer is on its second and third use."""
    out = str(tmp_path / 'seen.npy')
    mp.spawn(_overlap_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert np.load(out).tolist() == [[s * 10, s * 10 + 1] for s in range(5)]


def _checksum_worker(rank, world, port, out_path):
    for p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from training import distributed_render as dr
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    og = dr.OverlappedFrameGather([2, 4, 8, 3], torch.device('cpu'), rank, world, depth=2, checksums=6)
    g = torch.Generator().manual_seed(100 + rank)
    for step in range(9):                          # more submissions than the checksum ring keeps: only the last 6 can be asked for
        og.slot().copy_(torch.randint(0, 256, (2, 4, 8, 3), generator=g, dtype=torch.uint8))
        og.submit()
    og.drain()
    sent = og.sent_checksums(3, 6).contiguous()
    every = [torch.zeros_like(sent) for _ in range(world)]
    dist.all_gather(every, sent)
    if rank == 0:
        rc = og.received_checksums(3, 6)
        buf = og.received(8)[1].clone()
        a = int(og._sum64(buf))
        buf.view(-1)[5] ^= 1                       # one bit in one byte
        b = int(og._sum64(buf))
        swapped = og.received(8)[1].clone().view(-1)
        swapped[:8], swapped[8:16] = swapped[8:16].clone(), swapped[:8].clone()      # two 64-bit words exchanged: a plain sum would not see it
        c = int(og._sum64(swapped.view(2, 4, 8, 3)))
        np.save(out_path, np.array([int((rc != torch.stack(every, dim=1)).sum()), int(a == b), int(a == c and not torch.equal(swapped[:8], swapped[8:16])),
                                    int(len(set(sent.tolist())) == 6)]))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_frame_gather_checksums_cover_every_submission(tmp_path):
    """`checksums=K`: what arrived from every rank in each of the last K submissions has the checksum the sender formed; the checksum sees a
    flipped bit and two exchanged words (bench.py `gather_check.checksummed_steps`)."""
    out = str(tmp_path / 'sums.npy')
    mp.spawn(_checksum_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    mismatches, blind_to_bit, blind_to_swap, distinct = np.load(out).tolist()
    assert mismatches == 0 and blind_to_bit == 0 and blind_to_swap == 0 and distinct == 1
