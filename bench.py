"""bench.py — throughput of the IDE-3D render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one batch: `G.mapping` + `G.synthesis` (backbone -> tri-planes -> fused
ray-marcher at 96 samples -> 64->512 super-resolution, RGB + 19-class seg) for 4 seeds on every rank, conversion to
uint8 RGB|seg frames, and (N > 1) an RCCL gather of the uint8 frames to rank 0 — BASELINE.json config 2 per GPU,
config 4's sharding across GPUs.  Random-init ide3d-ffhq-64-512 generator, synthetic latents, fp32.
Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      the tri-plane gather kernel (the kernel BASELINE's metric names): algorithmic bytes / HIP-event time
  cpu_baseline  the CPU oracle ("port" of the reference's PyTorch CPU path) timed on the host cores, rank 0, N = 1 only
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12            # B/s, MI355X spec (MI355X_MICROARCH.md)
FP32_MFMA_PEAK = 157.3e12    # FLOP/s
BATCH = 4                    # seeds per rank per step (BASELINE config 2)
PROFILE_ROUND = 'round1'


def gather_bytes(n_images, C=32, H=256, W=256, M=64 * 64 * 96, triplanes=1):
    """SURVEY.md §8(d) canonical formula: planes + coords + output, fp32."""
    return (3 * C * H * W + 3 * M + C * M) * 4 * n_images * triplanes


def conv_flops(spec, n_images):
    """2 * Cin * Cout * k^2 * (positions the kernel is evaluated at) over the convs of one synthesis pass."""
    fl = 0
    res_list = spec.voxel_resolutions()
    pc = 3 * spec.plane_channels
    for i, res in enumerate(res_list):
        cout = spec.voxel_width(res)
        if i > 0:
            cin = spec.voxel_width(res // 2)
            fl += 2 * cin * cout * 9 * (res // 2) ** 2            # transposed conv evaluated on the low-res grid
        fl += 2 * cout * cout * 9 * res ** 2
        fl += 2 * cout * pc * res ** 2 * 2                        # torgb + toseg
    cin = spec.feature_channels
    widths = spec.sr_widths()
    for res in spec.sr_resolutions():
        cout = widths[res]
        fl += 2 * cin * cout * 9 * (res // 2) ** 2
        fl += 2 * cout * cout * 9 * res ** 2
        fl += 2 * cout * (spec.img_channels + spec.seg_channels) * res ** 2
        cin = cout
    return fl * n_images


def bench_gather(device, iters=20, tiled=True):
    """Isolated `sample_from_triplane` at the benchmark shape (N=4 images, channels_last planes): HIP events.
    tiled=True passes the ray-grid hint the renderer has (LDS-staged kernel); False times the flat kernel."""
    from dnnlib import util
    g = torch.Generator().manual_seed(0)
    n, C, H, M = BATCH, 32, 256, 64 * 64 * 96
    planes = torch.randn(n, 3 * C, H, H, generator=g).to(device).contiguous(memory_format=torch.channels_last)
    # coordinates of a real camera frustum (rays x depth steps), so locality matches the renderer's access pattern
    from training import triplane, volumetric_rendering as vr
    pts, z, d = vr.get_initial_rays_trig(n, 96, device, 18.0, (64, 64), 2.25, 3.3)
    cam = torch.cat([triplane.camera_label(y, device=device) for y in (-0.5, -0.15, 0.2, 0.5)])[:, :16].reshape(-1, 4, 4)
    wp, *_ = vr.transform_sampled_points(pts, z, d, device, h_stddev=0, v_stddev=0, camera=cam, mode=None,
                                         jitter=torch.rand(z.shape, generator=g).to(device))
    coords = (wp.reshape(n, M, 3) * float(os.environ.get('IDE3D_BENCH_COORD_SCALE', '1'))).contiguous()   # experiment knob
    del pts, z, d, wp
    ray_grid = (64, 64, 96) if tiled else None
    for _ in range(3):
        util.sample_from_triplane(coords, planes, ray_grid=ray_grid)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); util.sample_from_triplane(coords, planes, ray_grid=ray_grid); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    avg = sum(ms) / len(ms)
    algo = gather_bytes(n)
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same launch shape (FETCH_SIZE doubled per the
    # gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE); null when the summary is absent.
    traffic = None
    pmc = os.path.join(ROOT, 'profiles', PROFILE_ROUND, 'gather_tile_pmc.json' if tiled else 'gather_pmc.json')
    if os.path.isfile(pmc):
        traffic = json.load(open(pmc)).get('hbm_traffic_bytes_per_launch')
    return dict(kernel='triplane_sample_tile_kernel' if tiled else 'triplane_sample_cl2_kernel', bound='hbm', achieved=algo / (avg * 1e-3) / 1e9, peak=HBM_PEAK / 1e9, unit='GB/s',
                frac=algo / (avg * 1e-3) / HBM_PEAK, traffic=traffic, bytes_per_launch=algo, avg_launch_us=avg * 1e3, min_launch_us=ms[0] * 1e3,
                launch_shape=f'N={n} images x 1 tri-plane (C=32, 256x256), M=393216 samples/image')


def cpu_baseline(budget_s=20.0):
    """The CPU oracle (fp32 PyTorch-CPU port of the reference path) on the host cores: full-size generator, 1 seed per pass."""
    from oracle import fast_ops, generator as ogen, spec as ospec
    from training import triplane
    torch.manual_seed(0)
    G = triplane.TriPlaneGenerator().eval()
    sd = {k: v.detach() for k, v in G.state_dict().items()}
    sp = ospec.Spec()
    c = triplane.camera_label(0.0)
    cond = triplane.conditioning_label()
    # pick the thread count that is fastest on this host (a 256-thread pool is slower than 32 on small conv shapes)
    probe_ws = ogen.mapping(sd, sp, torch.zeros(1, 512), cond, ops=fast_ops)
    probe_feat = torch.randn(1, sp.feature_channels + sp.seg_channels, 64, 64)
    best, cores = None, 1
    for t in sorted({min(os.cpu_count() or 1, n) for n in (8, 16, 32, 64, 128, 256)}):
        torch.set_num_threads(t)
        ogen.superres(sd, sp, probe_feat, probe_ws, 'const', fast_ops)
        t0 = time.perf_counter(); ogen.superres(sd, sp, probe_feat, probe_ws, 'const', fast_ops); dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, t
    torch.set_num_threads(cores)

    def one(seed):
        z = torch.from_numpy(np.random.RandomState(seed).randn(1, 512))
        ws = ogen.mapping(sd, sp, z, cond, ops=fast_ops)
        jit = torch.rand(1, 4096, 96)
        return ogen.synthesis(sd, sp, ws, c, jitter=jit, ops=fast_ops)

    one(0)      # warm-up (allocator, thread pools)
    n, t0 = 0, time.perf_counter()
    while True:
        one(n + 1); n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 8:
            break
    return dict(value=n / dt, unit='frames/s', cores=cores, kind='port',
                sample=f'{n} full-size 512x512 RGB+seg frames (1 seed each, 96 samples, fp32 torch-CPU oracle, {cores} of {os.cpu_count()} host threads), {dt:.1f} s')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--graph', type=int, default=1, help='replay G.mapping + G.synthesis from a captured hipGraph (0 = eager launches)')
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Native libraries write banners to file descriptor 1 (RCCL prints its
    # version / host / library path at communicator creation), so fd 1 is pointed at stderr for the whole run and the
    # JSON line goes to a duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1 or os.environ.get('IDE3D_BENCH_FORCE_DIST'):     # the env knob exercises the RCCL path on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
    assert args.gpus == world, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    from torch_utils import hip_plugin
    from training import triplane
    from training import distributed_render as dr
    hip_plugin.load()     # hard error if the HIP library is missing

    torch.manual_seed(0)  # same random-init weights on every rank
    G = triplane.TriPlaneGenerator().eval().to(device)
    spec = G.spec
    cond = triplane.conditioning_label(device).repeat(BATCH, 1)
    yaws = [-0.5, 0.0, 0.5, 0.25]
    cams = torch.cat([triplane.camera_label(y, device=device) for y in yaws])
    palette = dr.palette_tensor(spec.seg_channels, device)
    gathered = [torch.empty([BATCH, 512, 1024, 3], dtype=torch.uint8, device=device) for _ in range(world)] if (dist and rank == 0) else None

    graphed = None
    if args.graph:
        try:
            graphed = triplane.GraphedRenderer(G, BATCH, device)
        except Exception as e:      # capture is an optimisation, never a requirement
            print(f'[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); running eagerly', file=sys.stderr)
            graphed = None

    def step(i):
        seeds = [(i * world + rank) * BATCH + j for j in range(BATCH)]
        z = torch.from_numpy(np.stack([np.random.RandomState(s).randn(512) for s in seeds])).to(device)
        with torch.no_grad():
            if graphed is not None:
                img, seg = graphed(z, cond, cams)
            else:
                ws = G.mapping(z, cond)
                img, seg = G.synthesis(ws, c=cams, noise_mode='const', return_seg=True)
            frames = dr.frames_u8(img, seg, palette)
        if dist:
            dist.gather(frames, gathered, dst=0)
        return frames

    for i in range(args.warmup):
        step(i)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)

    if rank == 0:
        frames_total = BATCH * world * args.steps
        out = {
            'metric': '512x512 RGB+seg frames/s @96 depth samples (whole job)', 'value': frames_total / dt, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'gen_images.py-style: random-init ide3d-ffhq-64-512, G.mapping + G.synthesis (64 neural render -> 512, '
                                   '96 samples, RGB + 19-class seg) + uint8 frame conversion, batch = 4 seeds per GPU',
                       'global_batch': BATCH * world, 'parallelism': f'dp{world} (one rank per GPU, RCCL gather of uint8 frames)'},
            'frames_per_s_per_gpu': frames_total / dt / world,
            'conv_tflops': conv_flops(spec, BATCH) * args.steps / dt / 1e12,
            'native_launches': dict(hip_plugin.CALLS), 'hip_graph': graphed is not None,
        }
        if not args.no_roofline:
            out['roofline'] = bench_gather(device)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
