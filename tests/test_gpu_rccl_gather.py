"""The sharded path's exchange step on the one GPU the driver has: `OverlappedFrameGather` on the `nccl` backend (= RCCL) at world size 1.
RCCL's stream runs beside the split-arithmetic convolutions of the captured render graph — the neighbour DESIGN.md section 4.2 is about — for
>= 200 overlapped steps per arithmetic; every received buffer is compared byte for byte with a blocking render of the same step's inputs,
and the per-submission checksums (what bench.py's `gather_check` relies on at N > 1) agree for every step.  `pytest -m gpu`."""

import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS = 208
BATCH = 4


@pytest.fixture(scope='module')
def rccl_world1(gpu_device):
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(gpu_device)
    dist.init_process_group(backend='nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=gpu_device)
    yield dist
    dist.destroy_process_group()


@pytest.fixture(scope='module')
def generator(gpu_device):
    from training import triplane
    torch.manual_seed(0)
    return triplane.TriPlaneGenerator().eval().requires_grad_(False).to(gpu_device)


@pytest.mark.parametrize('arith', ['bf16x6', 'f16x3'])
def test_overlapped_rccl_gather_beside_the_render_graph(rccl_world1, generator, gpu_device, arith):
    from training import distributed_render as dr
    from training import triplane
    G = generator
    res = G.img_resolution
    cond = triplane.conditioning_label(gpu_device).repeat(BATCH, 1)
    cams = torch.cat([triplane.camera_label(y, device=gpu_device) for y in (-0.5, 0.0, 0.5, 0.25)])
    palette = dr.palette_tensor(G.synthesis.seg_channels, gpu_device)
    run = triplane.GraphedRenderer(G, BATCH, gpu_device, conv_arithmetic=arith, static_labels=True)
    jit = torch.rand(BATCH, G.synthesis.render_size ** 2, G.spec.num_steps, generator=torch.Generator().manual_seed(5)).to(gpu_device)
    zs = torch.from_numpy(np.stack([np.random.RandomState(1000 + i).randn(BATCH, G.z_dim) for i in range(STEPS)])).float().to(gpu_device)

    # blocking reference: render, convert, keep — one step at a time, nothing in flight beside it
    want = torch.empty([STEPS, BATCH, res, 2 * res, 3], dtype=torch.uint8, device=gpu_device)
    with torch.no_grad():
        for i in range(STEPS):
            img, seg = run(zs[i], cond, cams, jitter=jit)
            dr.frames_u8(img, seg, palette, out=want[i])
            torch.cuda.synchronize()

    # overlapped: the gather of step k is in flight (RCCL's stream) while step k + 1 renders; buffers are reused every second step
    og = dr.OverlappedFrameGather([BATCH, res, 2 * res, 3], gpu_device, 0, 1, checksums=STEPS)
    got = torch.empty_like(want)
    with torch.no_grad():
        for i in range(STEPS):
            img, seg = run(zs[i], cond, cams, jitter=jit)
            dr.frames_u8(img, seg, palette, out=og.slot())
            assert og.submit() == i
            if i >= 1:
                og.wait(i - 1)                                   # stream-ordered; submission i stays in flight
                got[i - 1].copy_(og.received(i - 1)[0], non_blocking=True)
        og.drain()
        got[STEPS - 1].copy_(og.received(STEPS - 1)[0])
    torch.cuda.synchronize()
    bad = [i for i in range(STEPS) if not torch.equal(got[i], want[i])]
    assert not bad, f'{arith}: {len(bad)} of {STEPS} received buffers differ from the blocking render (first: step {bad[0]})'
    assert torch.equal(og.received_checksums(0, STEPS)[:, 0], og.sent_checksums(0, STEPS)), 'per-submission checksums'
    assert len(set(og.sent_checksums(0, STEPS).tolist())) == STEPS, 'every step rendered different frames'
