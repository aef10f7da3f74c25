"""bench.py — throughput of the IDE-3D render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With `--gpus N > 1` and no torch.distributed environment the script launches itself once per GPU
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per HIP device, RCCL);
started under an external `torch.distributed.run` it uses the RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* it is given.

A step = one pass of the hot path over one batch: `G.mapping` + `G.synthesis` (backbone -> tri-planes -> fused
ray-marcher at 96 samples -> 64->512 super-resolution, RGB + 19-class seg) for 4 seeds on every rank, conversion to
uint8 RGB|seg frames, and (N > 1) an RCCL gather of the uint8 frames to rank 0 — BASELINE.json config 2 per GPU,
config 4's sharding across GPUs.  Random-init ide3d-ffhq-64-512 generator, synthetic latents, fp32.

Timing: W untimed warm-up steps, then `--blocks` (default 5) timed blocks of EXACTLY K steps, each bracketed by a barrier +
`torch.cuda.synchronize()` on both sides, MAX over ranks per block; `value` / `ms_per_step` are the MEDIAN block (min / max
are reported next to it: box-to-box and run-to-run spread is a few per cent).

Rank 0 prints ONE COMPACT JSON line (< 4 KB, contract in the task statement; tests/test_bench_launcher_cpu.py checks the size and a
strict `json.loads`).  `value` is measured with the LIBRARY-DEFAULT convolution arithmetic (bf16x6: every fp32 operand as three bf16
pieces, six products, fp32 accumulation — as accurate as the fp32 matrix instruction against float64) — what a drop-in caller gets; exact
fp32 products (`value_fp32_exact`) and the other arithmetics are listed next to it in `by_conv_arithmetic`.  Extra objects in the line:
  roofline         the tri-plane gather kernel (the kernel BASELINE's metric names): algorithmic bytes / HIP-event time
  roofline_worst   the five hand-written kernels furthest below their roofline at their config-2 shapes (scripts/kernel_rooflines.py)
  cpu_baseline     the CPU oracle ("port" of the reference's PyTorch CPU path) timed on the host cores, rank 0, N = 1 only
  dropin_b1        the reference's gen_images.py:88-114 loop shape, loop body unchanged: batch 1, default arithmetic.  `frames_per_s` = the loop as it
                   runs (G.synthesis replays its own captured hipGraph), `eager_launches_frames_per_s` = capture switched off (rounds 1-5 called the
                   object `dropin_eager_b1`; renamed in round 6 because the key said eager and the number was not: ADVICE r5)
  parity_ok        after the timed region the SAME captured graph renders fixed inputs (seeds 0-3, fixed jitter) and the frames are
                   compared with the oracle fixture tests/golden/bench_parity.npz (oracle/make_bench_parity.py); with the
                   cpu_baseline leg on, the oracle frame computed there is compared live as well (`parity_live`)
The FULL record (every kernel's roofline row, the arithmetic sweep with timings, thread probes, notes) goes to
`gpurun_out/bench_full.json` and, pretty-printed, to stderr — never to stdout.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12            # B/s, MI355X spec (MI355X_MICROARCH.md)
HBM_COPY = 6.29e12           # B/s, measured copy ceiling (same guide)
FP32_MFMA_PEAK = 157.3e12    # FLOP/s
BATCH = 4                    # seeds per rank per step (BASELINE config 2)
PROFILE_ROUND = 'round6'
YAWS = (-0.5, 0.0, 0.5, 0.25)
PARITY_JITTER_SEED = 11      # = oracle/make_bench_parity.py
HEADLINE_ARITH = 'default'   # the library default (`ide3d_get_conv_arithmetic()` of a fresh process: bf16x6, fp32-grade) - what a drop-in caller runs
PARITY_TOL = 2e-5            # of the image / logit scale; measured 2-4e-6 in every arithmetic but bf16x3 (1.4e-5)
LINE_LIMIT = 4096            # bytes of the ONE stdout line (round 3's 20 KB line was not parsed by the driver)


# ---- launcher ----------------------------------------------------------------------------------------------------------------

def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, env=None, python=sys.executable, capture=False):
    """Run this script as `n` ranks of one node under torch.distributed.run (rendezvous on 127.0.0.1, a free port).
    Every rank inherits stdout, so rank 0's single JSON line is this process's output.  Returns the exit code
    (and stdout when `capture`)."""
    cmd = [python, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ if env is None else env)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: required by RCCL on this host driver
    env.setdefault('OMP_NUM_THREADS', '8')
    if capture:
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        return res.returncode, res.stdout
    return subprocess.run(cmd, env=env).returncode


# ---- per-kernel figures ---------------------------------------------------------------------------------------------------------

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


def bench_gather(device, iters=100, tiled=True, warm_launches=400):
    """Isolated `sample_from_triplane` at the benchmark shape (N=4 images, channels_last planes): HIP events around `iters`
    back-to-back launches, after `warm_launches` untimed ones.  The warm-up matters: the GPU's power management needs ~30 ms of
    continuous load to reach its sustained clocks (measured with scripts/micro/gather_bench: the same kernel takes 92 us in
    its first 30 launches after an idle period and 76.5 us from launch ~300 on) — inside the render pipeline the GPU is
    busy back to back, so the sustained figure is the representative one.
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
    for _ in range(max(3, warm_launches)):
        util.sample_from_triplane(coords, planes, ray_grid=ray_grid)
    # (1) the reported figure: `iters` launches back to back on the launch stream between ONE pair of HIP events, divided by `iters` - the
    #     kernel as the render pipeline runs it (the next dispatch overlaps the previous kernel's tail); includes the inter-launch gaps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        util.sample_from_triplane(coords, planes, ray_grid=ray_grid)
    e1.record()
    torch.cuda.synchronize()
    avg = e0.elapsed_time(e1) / iters
    # (2) next to it (full record only): an event pair around EVERY launch, as rounds 1-3 reported - each interval then also holds the
    #     dispatch latency behind an event packet (~3-4 us on this part), which no consumer of the kernel pays
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); util.sample_from_triplane(coords, planes, ray_grid=ray_grid); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    algo = gather_bytes(n)
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same launch shape (FETCH_SIZE doubled per the
    # gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE); null when the summary is absent.
    traffic, traffic_source = None, None
    for rnd in (PROFILE_ROUND, 'round5', 'round3', 'round2', 'round1'):
        pmc = os.path.join(ROOT, 'profiles', rnd, 'gather_tile_pmc.json' if tiled else 'gather_pmc.json')
        if os.path.isfile(pmc):
            rec = json.load(open(pmc))
            traffic = rec.get('hbm_traffic_bytes_per_launch')
            traffic_source = (f'profiles/{rnd}/{os.path.basename(pmc)} (kernel {rec.get("kernel", "?")}): rocprofv3 --pmc passes of this launch shape, '
                              'committed; NOT measured in this run')
            break
    rate = algo / (avg * 1e-3)
    kernel = 'triplane_sample_tile_pc_kernel' if tiled and os.environ.get('IDE3D_GATHER_PC', '8') != '0' else \
             'triplane_sample_tile_kernel' if tiled else 'triplane_sample_cl2_kernel'
    return dict(kernel=kernel, bound='hbm', achieved=rate / 1e9, peak=HBM_PEAK / 1e9,
                unit='GB/s', frac=rate / HBM_PEAK, frac_of_measured_copy_ceiling=rate / HBM_COPY, traffic=traffic, traffic_source=traffic_source, bytes_per_launch=algo,
                avg_launch_us=avg * 1e3, timing='one HIP event pair around `timed_launches` back-to-back launches / timed_launches',
                per_launch_event_pairs_us=dict(avg=sum(ms) / len(ms) * 1e3, min=ms[0] * 1e3, median=ms[len(ms) // 2] * 1e3),
                timed_launches=iters, warm_launches=warm_launches,
                launch_shape=f'N={n} images x 1 tri-plane (C=32, 256x256), M=393216 samples/image')


def live_pmc(passes, script_argv, kernel_substr, timeout_s=90, env_extra=None):
    """Counters of one kernel collected IN THIS RUN: one `rocprofv3 --pmc <counters>` pass per entry of `passes` (`--kernel-trace` is the only
    trace domain beside them) over `python <script_argv>` in a child process while this one is idle.  Returns ({counter: mean per launch},
    launches seen) or (None, reason) when the profiler is absent, fails or exceeds `timeout_s` per pass."""
    import csv, glob, shutil, signal, tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.isfile(exe):
        return None, 'rocprofv3 not found'
    if os.environ.get('ROCP_TOOL_LIBRARIES') or 'rocprofiler' in os.environ.get('LD_PRELOAD', ''):
        return None, 'this process is itself running under rocprofv3'
    tmp = tempfile.mkdtemp(prefix='ide3d_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp', **(env_extra or {}))
    vals, seen = {}, 0
    try:
        for i, ctrs in enumerate(passes):
            cmd = [exe, '--pmc', *ctrs, '--kernel-trace', '--output-format', 'csv', '-d', os.path.join(tmp, f'p{i}'), '--', sys.executable, *script_argv]
            pr = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)          # exactly the process group started here
                pr.wait()
                return None, f'rocprofv3 --pmc {" ".join(ctrs)} pass exceeded {timeout_s} s'
            got = {c: [] for c in ctrs}
            for f in glob.glob(os.path.join(tmp, f'p{i}', '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel_substr in r.get('Kernel_Name', '') and r.get('Counter_Name') in got:
                        got[r['Counter_Name']].append(float(r['Counter_Value']))
            if not all(got.values()):
                return None, f'rocprofv3 --pmc {" ".join(ctrs)} pass returned no rows for {kernel_substr} (exit code {pr.returncode})'
            for c, v in got.items():
                vals[c] = sum(v) / len(v)
                seen = max(seen, len(v))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return vals, seen


def live_gather_traffic(tiled=True):
    """HBM bytes per launch of the gather kernel, measured in this run: bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 (gfx950:
    FETCH_SIZE reports half of a 16-byte-per-lane streaming read, MI355X_MICROARCH.md "HBM"; the two do not fit one pass)."""
    vals, seen = live_pmc([('FETCH_SIZE',), ('WRITE_SIZE',)], [os.path.join(ROOT, 'scripts', 'gather_only.py'), '3', 'tile' if tiled else 'flat', '3'], 'triplane_sample')
    if vals is None:
        return None, seen
    return 2.0 * vals['FETCH_SIZE'] * 1024.0 + vals['WRITE_SIZE'] * 1024.0, (
        f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes run by this bench.py (mean over {seen} launches of the same shape): 2 * FETCH_SIZE KB + WRITE_SIZE KB')


def live_mfma_busy(kernel_name, arith):
    """Matrix-pipe busy share of the governing convolution kernel, measured in this run over scripts/modconv_only.py (the layers' own shapes):
    SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE) — busy cycles summed over 1024 SIMDs, GUI_ACTIVE over 8 XCDs (scripts/pmc_summarize.py)."""
    vals, seen = live_pmc([('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE')], [os.path.join(ROOT, 'scripts', 'modconv_only.py'), '2'], kernel_name,
                          env_extra={'IDE3D_CONV_ARITH': arith})
    if vals is None or not vals.get('GRBM_GUI_ACTIVE'):
        return None, seen
    return vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (128.0 * vals['GRBM_GUI_ACTIVE']), f'rocprofv3 --pmc pass run by this bench.py over scripts/modconv_only.py ({seen} launches of this kernel)'


# ---- CPU baseline --------------------------------------------------------------------------------------------------------------

def cpu_baseline(budget_s=20.0, parity_inputs=None):
    """The CPU oracle (fp32 PyTorch-CPU port of the reference path) on the host cores: full-size generator, 1 seed per pass.
    Returns (record, oracle output of parity image 0 or None)."""
    from oracle import fast_ops, generator as ogen, spec as ospec
    from training import triplane
    torch.manual_seed(0)
    G = triplane.TriPlaneGenerator().eval()
    sd = {k: v.detach() for k, v in G.state_dict().items()}
    sp = ospec.Spec()
    c = triplane.camera_label(0.0)
    cond = triplane.conditioning_label()

    def one(seed, cam=c, jit=None, zrow=None):
        z = torch.from_numpy(np.random.RandomState(seed).randn(1, 512)) if zrow is None else zrow
        ws = ogen.mapping(sd, sp, z, cond, ops=fast_ops)
        jit = torch.rand(1, 4096, 96) if jit is None else jit
        return ogen.synthesis(sd, sp, ws, cam, jitter=jit, ops=fast_ops)

    # thread count: probed on the WHOLE pass (mapping + backbone + renderer + super-resolution), one frame per candidate; a
    # 256-thread pool is slower than a few dozen threads on these conv shapes
    t_all = time.perf_counter()
    best, cores, probes = None, 1, {}
    for t in sorted({min(os.cpu_count() or 1, n) for n in (8, 16, 32, 64)}):
        torch.set_num_threads(t)
        t0 = time.perf_counter(); one(100 + t); dt = time.perf_counter() - t0
        probes[t] = round(dt, 2)
        if best is None or dt < best:
            best, cores = dt, t
        if time.perf_counter() - t_all > budget_s:
            break
    torch.set_num_threads(cores)
    live = None
    n, t0 = 0, time.perf_counter()
    if parity_inputs is not None:      # the first timed frame doubles as the live parity reference (same work as any other frame)
        z, cams, _cond, jit = parity_inputs
        live = one(0, cam=cams[:1], jit=jit[:1], zrow=z[:1]); n = 1
    while True:
        dt = time.perf_counter() - t0
        if (n >= 1 and dt > budget_s) or n >= 8:
            break
        one(n + 1); n += 1
    dt = time.perf_counter() - t0
    rec = dict(value=n / dt, unit='frames/s', cores=cores, kind='port',
               note='fp32 torch-CPU restatement of the reference path (oracle/fast_ops.py + generator.py), golden-pinned to outputs of '
                    '/root/reference run in the build container (tests/test_oracle_golden.py); /root/reference itself does not exist on this box',
               thread_probe_s_per_frame=probes,
               sample=f'{n} full-size 512x512 RGB+seg frames (1 seed each, 96 samples, fp32 torch-CPU oracle, {cores} of {os.cpu_count()} host threads), {dt:.1f} s')
    return rec, live


# ---- parity of the benchmarked graph -----------------------------------------------------------------------------------------------

def parity_inputs():
    """= oracle/make_bench_parity.parity_inputs (restated: the timed script never imports oracle/)."""
    from training import triplane
    z = torch.from_numpy(np.stack([np.random.RandomState(s).randn(512) for s in range(BATCH)]))
    cams = torch.cat([triplane.camera_label(y) for y in YAWS])
    cond = triplane.conditioning_label().repeat(BATCH, 1)
    jit = torch.rand(BATCH, 4096, 96, generator=torch.Generator().manual_seed(PARITY_JITTER_SEED))
    return z, cams, cond, jit


def check_parity(render, device, tol=PARITY_TOL):
    """Render the fixed parity inputs through `render` (the benchmarked callable) and compare with the oracle fixture."""
    path = os.path.join(ROOT, 'tests', 'golden', 'bench_parity.npz')
    z, cams, cond, jit = parity_inputs()
    img, seg = render(z.to(device), cond.to(device), cams.to(device), jit.to(device))
    img, seg = img.float().cpu(), seg.float().cpu()
    rec = dict(fixture='tests/golden/bench_parity.npz (CPU oracle, oracle/make_bench_parity.py)', tol_rel=tol)
    if not os.path.isfile(path):
        rec.update(ok=None, error='fixture missing')
        return rec, (img, seg)
    d = np.load(path)
    errs = {'img': float((img[:, :, ::8, ::8] - torch.from_numpy(d['img'])).abs().max()) / float(d['scale_img']),
            'seg': float((seg[:, :, ::16, ::16] - torch.from_numpy(d['seg'])).abs().max()) / float(d['scale_seg'])}
    rec.update(ok=bool(max(errs.values()) <= tol), max_rel_err=errs, images=int(img.shape[0]))
    return rec, (img, seg)


def _finite(o):
    """NaN / +-inf -> None (strict JSON has neither), recursively."""
    if isinstance(o, float):
        return o if o == o and abs(o) != float('inf') else None
    if isinstance(o, dict):
        return {str(k): _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, (np.floating, np.integer)):
        return _finite(o.item())
    return o


OPTIONAL_KEYS = ('gpu_state', 'roofline_worst', 'dropin_b1', 'parity_live', 'gather_overlap', 'frames_per_s_by_rank', 'timing', 'by_conv_arithmetic', 'parity', 'roofline_step')


def gpu_state_under_load(enqueue, busy_steps=300):
    """Clock / power of GPU 0 WHILE the render step runs: `busy_steps` steps are enqueued (asynchronous graph replays: ~1.3 s of GPU work), then
    `rocm-smi --showclocks --showpower --json` is read on the host while they execute.  The boxes of the pool differ by up to 10 % under bf16 matrix
    load with the same binary: this puts the clock the box actually ran at next to every number."""
    import shutil
    import subprocess
    exe = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
    if not os.path.exists(exe):
        return None
    try:
        for i in range(busy_steps):
            enqueue(i)
        t0 = time.perf_counter()
        res = subprocess.run([exe, '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=20)
        took = time.perf_counter() - t0
        torch.cuda.synchronize()
        still_busy = time.perf_counter() - t0 - took > 0.02          # the queue outlived the query: it was sampled under load
        card = next(iter(json.loads(res.stdout).values()))
        pick = {}
        for k, v in card.items():
            kl = k.lower()
            if ('sclk' in kl or 'mclk' in kl) and 'speed' in kl:
                pick[kl.split()[0] + '_mhz'] = v.strip('()').replace('Mhz', '').replace('MHz', '')
            elif 'power' in kl and '(w)' in kl:
                pick['power_w'] = v
        pick['sampled_under_load'] = bool(still_busy)
        return pick
    except Exception as e:      # noqa: BLE001 - diagnostics only
        return {'error': f'{type(e).__name__}: {e}'[:120]}


def compact_line(out, limit=None):
    """The ONE stdout line: strict JSON (ASCII, no NaN / Infinity), at most LINE_LIMIT bytes.  Should it ever grow past the limit, the
    optional objects are dropped one by one (the contract keys, `roofline` and `cpu_baseline` stay) — round 3's 20 KB line was not parsed."""
    limit = LINE_LIMIT if limit is None else limit
    out = _finite(dict(out))
    line = json.dumps(out, allow_nan=False, ensure_ascii=True, separators=(',', ':'))
    dropped = []
    for k in OPTIONAL_KEYS:
        if len(line) <= limit:
            break
        if k in out:
            out.pop(k); dropped.append(k)
            out['dropped_for_size'] = dropped
            line = json.dumps(out, allow_nan=False, ensure_ascii=True, separators=(',', ':'))
    assert len(line) <= limit, f'bench line is {len(line)} bytes'
    return line


def dropin_b1(G, device, palette, images=24, warm=4):
    """What a drop-in caller of the reference's gen_images.py:88-114 gets: one seed at a time (batch 1), the loop body UNCHANGED — host
    latents per seed, `G.mapping` -> `G.synthesis(return_seg=True)` -> uint8 RGB | coloured seg frame — in the library-default arithmetic.
    Since round 5 `G.synthesis` itself captures a repeated call signature into a hipGraph and replays it (training/graph_cache.py):
    `frames_per_s` is that loop as it runs; `eager_launches_frames_per_s` the same loop with the capture switched off (rounds 1-4).
    Timed over `images` seeds between two synchronisations."""
    from training import triplane
    from training import graph_cache
    from training import distributed_render as dr
    cond = triplane.conditioning_label(device)
    cam = triplane.camera_label(0.0, device=device)

    def one(seed):
        z = torch.from_numpy(np.random.RandomState(seed).randn(1, G.z_dim)).to(device).float()
        with torch.no_grad():
            ws = G.mapping(z, cond)
            img, seg = G.synthesis(ws, c=cam, noise_mode='const', return_seg=True)
            return dr.frames_u8(img, seg, palette)

    def timed():
        for s in range(warm):
            one(10_000 + s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(images):
            one(s)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    before = dict(graph_cache.STATS)
    dt = timed()
    stats = {k: graph_cache.STATS[k] - before.get(k, 0) for k in ('eager', 'capture', 'replay')}
    rec = {'frames_per_s': round(images / dt, 1), 'ms_per_image': round(dt / images * 1e3, 3), 'images': images,
           'what': 'gen_images.py:88-114 loop shape, loop body unchanged: batch 1, library-default arithmetic; G.synthesis replays its own captured hipGraph',
           'synthesis_calls': stats}
    with graph_cache.disabled():
        dt = timed()
    rec['eager_launches_frames_per_s'] = round(images / dt, 1)
    # the same loop with mapping AND synthesis inside one batch-1 hipGraph and pinned latents (`triplane.GraphedRenderer(G, 1, device)`)
    run = triplane.GraphedRenderer(G, 1, device, static_labels=True)

    def one_graphed(seed):
        z = torch.from_numpy(np.random.RandomState(seed).randn(1, G.z_dim)).float().pin_memory()
        img, seg = run(z, cond, cam)
        with torch.no_grad():
            return dr.frames_u8(img, seg, palette)

    for s in range(warm):
        one_graphed(20_000 + s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(images):
        one_graphed(s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec['graphed_b1_frames_per_s'] = round(images / dt, 1)
    return rec


ARITH_DTYPE_SHORT = {'fp32': 'f32', 'f16x3': 'f32 (3x3 convs: f16x3 split, f32 accumulate)', 'bf16x6': 'f32 (3x3 convs + MLPs: fp32 operands as 3 bf16 pieces, 6 products, f32 accumulate)',
                     'bf16x3': 'f32 (3x3 convs: bf16x3 split, f32 accumulate)'}
ARITH_DTYPE = {'fp32': 'f32', 'f16x3': 'f32 (3x3 convolutions: 2-way fp16 split operands with exact power-of-two range scales, 3 products ~2^-21, fp32 accumulate; all else f32)',
               'bf16x6': 'f32 (3x3 convolutions: 3-way bf16 split operands, 6 products >= 2^-24, fp32 accumulate; all else f32)',
               'bf16x3': 'f32 storage, 3x3 convolutions bf16x3 (2-way split operands, 3 products, ~2^-17 per product, fp32 accumulate)'}


# ---- main ------------------------------------------------------------------------------------------------------------------------

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--blocks', type=int, default=5, help='timed blocks of --steps steps each; the median block is reported')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-roofline-extra', action='store_true')
    ap.add_argument('--no-live-pmc', action='store_true', help='take roofline.traffic from the committed PMC summary instead of two rocprofv3 --pmc passes in this run')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-dropin', action='store_true', help='skip the batch-1 eager leg (gen_images.py loop shape)')
    ap.add_argument('--graph', type=int, default=1, help='replay G.mapping + G.synthesis from a captured hipGraph (0 = eager launches)')
    ap.add_argument('--conv-arith', default=HEADLINE_ARITH, choices=['default', 'fp32', 'bf16x6', 'f16x3', 'bf16x3'],
                    help='arithmetic of the shared-weight 3x3 convolutions (include/ide3d_hip.h).  "default" = the library default (bf16x6: fp32-grade '
                         'split products on the bf16 matrix pipe), which is what the headline is measured with; fp32 = exact fp32 products '
                         '(reported as value_fp32_exact either way)')
    ap.add_argument('--no-arith-sweep', action='store_true', help='skip the short runs with the other conv arithmetics (N = 1 only)')
    ap.add_argument('--blocking-gather', action='store_true', help='N > 1: synchronous gather on the compute stream in the timed steps')
    ap.add_argument('--dry-run-cpu', action='store_true',
                    help='launcher / protocol self-test without a GPU: tiny generator on CPU tensors, gloo backend (tests/test_bench_launcher_cpu.py)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher, one rank per GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # The contract is ONE JSON line on stdout.  Native libraries write banners to file descriptor 1 (RCCL prints its
    # version / host / library path at communicator creation), so fd 1 is pointed at stderr for the whole run and the
    # JSON line goes to a duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    cpu = args.dry_run_cpu
    dist = None
    if world > 1 or os.environ.get('IDE3D_BENCH_FORCE_DIST'):     # the env knob exercises the RCCL path on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        if cpu:
            dist.init_process_group(backend='gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
    assert args.gpus == world, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if cpu:
        device = torch.device('cpu')
        torch.set_num_threads(2)
    else:
        assert torch.cuda.is_available(), 'bench.py needs a GPU'
        assert local_rank < torch.cuda.device_count(), f'rank {rank}: HIP device {local_rank} of {torch.cuda.device_count()} does not exist'
        device = torch.device('cuda', local_rank)
        torch.cuda.set_device(device)
    sync = (lambda: None) if cpu else torch.cuda.synchronize

    from training import triplane
    from training import distributed_render as dr
    hip_plugin = None
    if not cpu:
        from torch_utils import hip_plugin
        hip_plugin.load()     # hard error if the HIP library is missing
        if args.conv_arith != 'default':
            hip_plugin.conv_arithmetic(args.conv_arith)
    arith = hip_plugin.conv_arithmetic() if hip_plugin else 'fp32'
    forced_blocking = False      # (round 3 forced the blocking gather beside split arithmetics; exclusive residency made that unnecessary: DESIGN.md 4.2)

    torch.manual_seed(0)  # same random-init weights on every rank
    G = triplane.TriPlaneGenerator(triplane.tiny_spec() if cpu else None).eval().to(device)
    spec = G.spec
    res = spec.img_resolution
    cond = triplane.conditioning_label(device).repeat(BATCH, 1)
    cams = torch.cat([triplane.camera_label(y, device=device) for y in YAWS])
    palette = dr.palette_tensor(spec.seg_channels, device)
    gathered = [torch.empty([BATCH, res, 2 * res, 3], dtype=torch.uint8, device=device) for _ in range(world)] if (dist and rank == 0) else None

    # The headline is the hipGraph replay.  A failed capture is an error (exit code != 0, no JSON line) unless eager launches were
    # asked for with --graph 0: a silently slower eager number must never stand in for the graph's.
    graphed = None
    if args.graph and not cpu:
        graphed = triplane.GraphedRenderer(G, BATCH, device, static_labels=True)      # cond / cams are fixed device tensors here

    def render(z, c_cond, c_cam, jitter=None):
        """The benchmarked callable: new latents (and fresh stratified jitter unless given) -> (img, seg)."""
        with torch.no_grad():
            if graphed is not None:
                return graphed(z, c_cond, c_cam, jitter=jitter)
            ws = G.mapping(z.float(), c_cond)
            return G.synthesis(ws, c=c_cam, noise_mode='const', return_seg=True, ray_jitter=jitter)

    # N > 1: the uint8 frames of step k travel to rank 0 (RCCL gather, 7 concurrent peer -> root xGMI copies) WHILE step k + 1 renders:
    # double-buffered send / receive tensors, `dist.gather(async_op=True)` on the communication stream (dr.OverlappedFrameGather).
    # `--blocking-gather` restores the synchronous gather on the compute stream (measured next to it as `ms_per_step_blocking_gather`).
    # `checksums`: every buffer sent / received in the timed blocks gets a position-weighted 64-bit checksum (sums of received buffers on a side
    # stream); after the timed region rank 0 compares what arrived from every rank in EVERY timed step with what that rank says it sent
    total_subs = max(1, args.blocks) * args.steps
    og = dr.OverlappedFrameGather([BATCH, res, 2 * res, 3], device, rank, world, checksums=total_subs + 8) if dist else None

    def latents(i, r=None):
        seeds = [(i * world + (rank if r is None else r)) * BATCH + j for j in range(BATCH)]
        z = torch.from_numpy(np.stack([np.random.RandomState(s).randn(spec.z_dim) for s in seeds]))
        # pinned + non_blocking: a pageable host-to-device copy blocks the host until the stream has drained, i.e. until the previous
        # step has finished, and the GPU then idles for the ~0.1 ms the host needs to enqueue the next step (scripts/step_timeline.py).
        # The graphed renderer copies a host tensor straight into its static input buffer (one launch instead of copy + conversion).
        if cpu:
            return z.to(device)
        z = z.float().pin_memory()
        return z if graphed is not None else z.to(device, non_blocking=True)

    def step(i, blocking=False, jitter=None):
        img, seg = render(latents(i), cond, cams, jitter)
        with torch.no_grad():
            if dist and not blocking:
                frames = dr.frames_u8(img, seg, palette, out=og.slot())
                og.submit()
            else:
                frames = dr.frames_u8(img, seg, palette)
                if dist:
                    dist.gather(frames, gathered, dst=0)
        return frames

    def barrier():
        if dist:
            og.drain()
            dist.barrier()
        sync()

    # The roofline kernel of BASELINE's metric (the isolated tri-plane gather) is measured first thing, on the GPU as the job finds it, and once
    # more after everything else; see below where the line is assembled.
    rf_first = bench_gather(device) if (not cpu and rank == 0 and not args.no_roofline) else None

    for i in range(args.warmup):
        step(i, blocking=args.blocking_gather)
    block_s, rank_s = [], []
    done = args.warmup
    first_timed_sub = og.submitted if (og is not None and not args.blocking_gather) else None
    for _b in range(max(1, args.blocks)):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(done + i, blocking=args.blocking_gather)
        barrier()
        dt = time.perf_counter() - t0
        done += args.steps
        mine = dt
        if dist:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        block_s.append(dt); rank_s.append(mine)

    # N > 1 extras: per-rank frame rates (own clock, median block) and the cost of the RCCL gather alone
    per_rank, gather_ms, blocking_ms, gather_check = None, None, None, None
    sums_bad = None
    if dist and first_timed_sub is not None and og.submitted - first_timed_sub == total_subs:
        barrier()
        sent = og.sent_checksums(first_timed_sub, total_subs).contiguous()
        every = [torch.zeros_like(sent) for _ in range(world)]
        dist.all_gather(every, sent)
        if rank == 0:
            sums_bad = int((og.received_checksums(first_timed_sub, total_subs) != torch.stack(every, dim=1)).sum())
    if dist:
        mine = torch.tensor([BATCH * args.steps / sorted(rank_s)[len(rank_s) // 2]], device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(t) for t in allr]
        # the same K steps with the gather synchronous on the compute stream (what round 2 timed), for comparison
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(done + i, blocking=True)
        barrier()
        tt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        blocking_ms = float(tt) / args.steps * 1e3
        done += args.steps
        # the gather alone
        frames = step(done, blocking=True); done += 1
        barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            dist.gather(frames, gathered, dst=0)
        barrier()
        gather_ms = (time.perf_counter() - t0) / 5 * 1e3
        # contents and order through the double buffer: three overlapped steps with fixed jitter; rank 0 renders every rank's frames
        # itself and compares them with what arrived (byte-equal: same weights, same inputs, same kernels on every rank)
        jit_fix = torch.rand(BATCH, spec.render_size ** 2, spec.num_steps, generator=torch.Generator().manual_seed(PARITY_JITTER_SEED)).to(device)
        subs = []
        for k in range(3):
            step(done + k, jitter=jit_fix)
            subs.append(og.submitted - 1)
            if rank == 0 and k >= 1:           # read submission k - 1 while k is in flight (its buffers are not reused before k + 1)
                og.wait(subs[k - 1]); sync()
                subs[k - 1] = [t.clone() for t in og.received(subs[k - 1])]
        barrier()
        if rank == 0:
            subs[2] = [t.clone() for t in og.received(subs[2])]
            bad = 0
            for k in range(3):
                for r in range(world):
                    img, seg = render(latents(done + k, r), cond, cams, jit_fix)
                    with torch.no_grad():
                        want = dr.frames_u8(img, seg, palette)
                    bad += int((subs[k][r] != want).any())
            gather_check = {'ok': bad == 0 and not sums_bad, 'steps': 3, 'ranks': world, 'mismatching_buffers': bad,
                            'what': 'frames received through the double-buffered async gather == frames rank 0 renders for the same (step, rank) inputs; '
                                    'and, for EVERY timed step, the 64-bit checksum of what arrived from each rank == the checksum that rank formed of what it sent',
                            'checksummed_steps': None if sums_bad is None else total_subs, 'checksum_mismatches': sums_bad}
        done += 3

    if rank == 0:
        order = sorted(block_s)
        med = order[len(order) // 2]
        frames_block = BATCH * world * args.steps
        r3 = lambda v, nd=3: None if v is None else round(float(v), nd)
        # `out` is the ONE stdout line (compact); `full` collects everything else for gpurun_out/bench_full.json + stderr
        out = {
            'metric': '512x512 RGB+seg frames/s @96 depth samples (whole job)', 'value': r3(frames_block / med, 2), 'unit': 'frames/s',
            'value_fp32_exact': r3(frames_block / med, 2) if arith == 'fp32' else None,     # else filled from the fp32 leg of the sweep below
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': r3(med / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': ARITH_DTYPE_SHORT[arith], 'data': 'synthetic',
            'config': {'workload': 'BASELINE config 2: random-init ide3d-ffhq-64-512, G.mapping + G.synthesis (64 render -> 512, 96 samples, RGB + 19-class seg) '
                                   '+ uint8 frames, batch 4 seeds per GPU',
                       'global_batch': BATCH * world, 'parallelism': f'dp{world}', 'conv_arithmetic': arith,
                       'conv_arithmetic_is_library_default': bool(args.conv_arith == 'default')},
            'timing': {'blocks': len(block_s), 'reported': 'median block', 'ms_per_step_min': r3(order[0] / args.steps * 1e3, 4),
                       'ms_per_step_max': r3(order[-1] / args.steps * 1e3, 4)},
            'conv_tflops': r3(conv_flops(spec, BATCH) * args.steps / med / 1e12, 1),
            'hip_graph': graphed is not None,
        }
        full = {'dtype_long': ARITH_DTYPE[arith], 'native_launches': dict(hip_plugin.CALLS) if hip_plugin else {},
                'timing_blocks_s': block_s, 'frames_per_s_per_gpu': frames_block / med / world}
        if dist:
            out['rccl_ranks'] = world
            out['frames_per_s_by_rank'] = [r3(v, 1) for v in per_rank]
            out['gather_ms'] = r3(gather_ms)
            out['gather_overlap'] = {'mode': 'blocking (split arithmetic: DESIGN.md 4.2)' if forced_blocking else
                                             'blocking' if args.blocking_gather else 'async double-buffered',
                                     'ms_per_step_overlapped': r3(med / args.steps * 1e3, 4), 'ms_per_step_blocking_gather': r3(blocking_ms, 4),
                                     'gather_ms_alone': r3(gather_ms)}
            out['gather_check'] = gather_check
            out['gather_bytes_per_rank_per_step'] = BATCH * res * 2 * res * 3
        if cpu:
            out['metric'] = 'DRY RUN (CPU tensors, tiny generator, gloo): launcher / protocol self-test, not a measurement'
            out['data'] = 'synthetic (cpu dry run)'
        if not cpu and world == 1:
            state = gpu_state_under_load(lambda i: step(i))
            full['gpu_state_under_load'] = state
            if state and 'error' not in state:
                out['gpu_state'] = {k: state[k] for k in list(state)[:5]}
        pin = None
        if not cpu and not args.no_parity:
            par, got = check_parity(render, device)
            full['parity'] = par
            out['parity_ok'] = par.get('ok')
            out['parity'] = {'max_rel_err': {k: float(f'{v:.3g}') for k, v in (par.get('max_rel_err') or {}).items()}, 'tol_rel': par['tol_rel'],
                             'vs': 'tests/golden/bench_parity.npz (CPU oracle)'}
            pin = parity_inputs()
        if not cpu and world == 1 and not args.no_dropin:
            out['dropin_b1'] = dropin_b1(G, device, palette)
        if not cpu and world == 1 and not args.no_arith_sweep:
            # the same step with the other arithmetics of the 3x3 layers: one captured graph each, the headline's timing protocol, parity against
            # the same golden frames (nothing else changes: every other kernel is fp32 in all of them)
            sweep_full = {arith: {'frames_per_s': frames_block / med, 'ms_per_step': med / args.steps * 1e3, 'blocks': len(block_s), 'steps_per_block': args.steps,
                                  'reported': 'median block', 'parity_ok': out.get('parity_ok'), 'parity_max_rel_err': (full.get('parity') or {}).get('max_rel_err')}}
            keep = graphed
            for other in ('fp32', 'bf16x6', 'f16x3', 'bf16x3'):
                if other == arith:
                    continue
                hip_plugin.conv_arithmetic(other)
                try:
                    graphed = triplane.GraphedRenderer(G, BATCH, device, static_labels=True) if keep is not None else None
                    for i in range(args.warmup):
                        step(done + i)
                    ts = []
                    for _b in range(max(1, args.blocks)):          # the protocol of the headline: K steps between synchronisations, median block
                        sync(); t0 = time.perf_counter()
                        for i in range(args.steps):
                            step(done + i)
                        sync(); ts.append(time.perf_counter() - t0)
                    med_o = sorted(ts)[len(ts) // 2]
                    rec = {'frames_per_s': BATCH * args.steps / med_o, 'ms_per_step': med_o / args.steps * 1e3, 'blocks': len(ts), 'steps_per_block': args.steps,
                           'reported': 'median block'}
                    if not args.no_parity:
                        par = check_parity(render, device)[0]
                        rec['parity_ok'] = par.get('ok'); rec['parity_max_rel_err'] = par.get('max_rel_err')
                    sweep_full[other] = rec
                finally:
                    graphed = keep
                    hip_plugin.conv_arithmetic(args.conv_arith)
            full['by_conv_arithmetic'] = sweep_full
            out['by_conv_arithmetic'] = {k: {'frames_per_s': r3(v['frames_per_s'], 1), 'parity_ok': v.get('parity_ok')} for k, v in sweep_full.items()}
            out['value_fp32_exact'] = r3(sweep_full['fp32']['frames_per_s'], 2)      # exact-fp32 products (v_mfma_f32_32x32x2_f32) in every convolution
        if not cpu and not args.no_roofline:
            # two measurements of the same kernel in one run (first thing / after everything else): their spread is the box's clock state, not
            # the kernel.  `roofline` is their MEAN — a fixed estimator (round 5 reported the faster one: min-of-N on the headline figure, ADVICE
            # r5 / VERDICT r5 #4) — and both travel in the line.
            rf_late = bench_gather(device)
            both = {'at_start': {'avg_launch_us': r3(rf_first['avg_launch_us'], 2), 'frac': r3(rf_first['frac'], 4)},
                    'after_sustained_load': {'avg_launch_us': r3(rf_late['avg_launch_us'], 2), 'frac': r3(rf_late['frac'], 4)}}
            rf = dict(rf_late)
            rf['avg_launch_us'] = 0.5 * (rf_first['avg_launch_us'] + rf_late['avg_launch_us'])
            rf['achieved'] = rf['bytes_per_launch'] / (rf['avg_launch_us'] * 1e-6) / 1e9
            rf['frac'] = rf['achieved'] / rf['peak']
            rf['timed_launches'] = rf_first['timed_launches'] + rf_late['timed_launches']
            full['roofline'] = rf
            full['roofline_both'] = {'at_start': rf_first, 'after_sustained_load': rf_late}
            live, why = (None, 'disabled (--no-live-pmc)') if (args.no_live_pmc or world != 1) else live_gather_traffic()
            if live is not None:
                rf['traffic_committed'], rf['traffic'], rf['traffic_source'] = rf['traffic'], live, why
            else:
                rf['traffic_live_failed'] = why
            out['roofline'] = {'kernel': rf['kernel'], 'bound': 'hbm', 'achieved': r3(rf['achieved'], 1), 'peak': rf['peak'], 'unit': 'GB/s',
                               'frac': r3(rf['frac'], 4), 'traffic': (int(rf['traffic']) if rf['traffic'] is not None else None), 'traffic_measured_in_this_run': live is not None,
                               'bytes_per_launch': rf['bytes_per_launch'], 'avg_launch_us': r3(rf['avg_launch_us'], 2), 'timed_launches': rf['timed_launches'],
                               'measured': 'twice in this run (first thing, and after everything else), 400 warm launches each; the MEAN of the two launch times is reported', **both}
        if not cpu and world == 1 and not args.no_roofline_extra:
            try:
                sys.path.insert(0, os.path.join(ROOT, 'scripts'))
                import kernel_rooflines
                extra = kernel_rooflines.measure_all(device, iters=10)
                full['roofline_extra'] = extra
                rows = [r for r in (extra.get('rows') if isinstance(extra, dict) else extra) or [] if isinstance(r, dict) and r.get('frac') is not None]
                rows.sort(key=lambda r: r['frac'])
                out['roofline_worst'] = [{'kernel': str(r.get('name', r.get('kernel', '?')))[:60], 'bound': r.get('bound'), 'frac': r3(r['frac'], 3),
                                          'us': r3(r.get('us', r.get('avg_us')), 1)} for r in rows[:5]]
                out['roofline_rows'] = len(rows)
                # the kernel that governs the frame rate (not only the isolated gather the metric names): the shared-weight 3x3 stride-1
                # convolutions of the 64^2 .. 256^2 layers — four launches of modconv_split_kernel<0,1,16,...> per step in the default arithmetic
                dom = [r for r in rows if r.get('kernel') == 'modconv_split_kernel' and f'[{arith}]' in r.get('name', '')
                       and any(t in r['name'] for t in ('3x3 512->512 @64', '3x3 256->256 @128', '3x3 128->128 @256'))]
                if dom and arith in ('bf16x6', 'f16x3', 'bf16x3'):
                    per_step = {'3x3 512->512 @64': 1, '3x3 256->256 @128': 1, '3x3 128->128 @256': 2}      # vb64.conv1, vb128.conv1, vb256.conv1 + b256.conv1
                    us = sum(r['us'] * n for r in dom for t, n in per_step.items() if t in r['name'])
                    fl = sum(r['algorithmic_flops'] * n for r in dom for t, n in per_step.items() if t in r['name'])
                    peak = dom[0]['peak']
                    pm = kernel_rooflines._pmc_notes()
                    # template arguments <MODE, BIG, tile rows, PARTS, weight buffers, waves, F16>: bf16x6 = 3 pieces, f16x3 = 2 fp16 pieces, bf16x3 = 2 bf16 pieces
                    kname = {'bf16x6': 'modconv_split_kernel<0, 1, 16, 3, 2, 8, 0>', 'f16x3': 'modconv_split_kernel<0, 1, 16, 2, 2, 8, 1>', 'bf16x3': 'modconv_split_kernel<0, 1, 16, 2, 2, 8, 0>'}[arith]
                    busy = pm.get(kname)
                    busy_live, busy_why = (None, 'disabled (--no-live-pmc)') if args.no_live_pmc else live_mfma_busy(kname, arith)
                    full['roofline_step_pmc'] = {'mfma_busy_frac_live': busy_live, 'source': busy_why, 'committed': busy}
                    out['roofline_step'] = {'kernel': kname.replace(', ', ',') + ' (3x3 stride-1, 64^2..256^2 layers)', 'bound': f'mfma:{arith}', 'launches_per_step': 4,
                                            'us_per_step': r3(us, 1), 'share_of_step': r3(us / (med / args.steps * 1e6), 3), 'achieved': r3(fl / us / 1e6, 1), 'peak': r3(peak, 1),
                                            'unit': 'TFLOP/s', 'frac': r3(fl / us / 1e6 / peak, 3),
                                            'mfma_busy_frac': (r3(busy_live, 3) if busy_live is not None else None if busy is None else busy['mfma_busy_frac']),
                                            'mfma_busy_measured_in_this_run': busy_live is not None}
            except Exception as e:
                out['roofline_worst'] = {'error': f'{type(e).__name__}: {e}'[:200]}
        if not cpu and world == 1 and not args.no_cpu_baseline:
            cb, live = cpu_baseline(parity_inputs=pin)
            full['cpu_baseline'] = cb
            out['cpu_baseline'] = {k: (r3(cb[k]) if k == 'value' else cb[k]) for k in ('value', 'unit', 'cores', 'kind', 'sample')}
            if live is not None and pin is not None:
                img, seg = got
                e_img = float((img[:1] - live['image']).abs().max()) / float(live['image'].abs().max())
                e_seg = float((seg[:1] - live['image_seg']).abs().max()) / float(live['image_seg'].abs().max())
                out['parity_live'] = {'vs': 'oracle frame computed in this run (image 0, full frame)', 'max_rel_err': {'img': float(f'{e_img:.3g}'), 'seg': float(f'{e_seg:.3g}')},
                                      'tol_rel': PARITY_TOL, 'ok': bool(max(e_img, e_seg) <= PARITY_TOL)}
        line = compact_line(out)
        full.update(line=json.loads(line))
        try:
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            with open(os.path.join(ROOT, 'gpurun_out', 'bench_full.json'), 'w') as f:
                json.dump(full, f, indent=1, default=str)
            out_path = 'gpurun_out/bench_full.json'
        except OSError:
            out_path = None
        if not cpu:
            sys.stderr.write('[bench.py] full record' + (f' ({out_path})' if out_path else '') + ':\n' + json.dumps(full, indent=1, default=str) + '\n')
        os.write(json_fd, (line + '\n').encode())
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
