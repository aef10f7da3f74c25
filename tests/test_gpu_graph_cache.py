"""`training/graph_cache.py`: the hipGraph that `G.synthesis` captures on the caller's behalf replays exactly what the eager launches
compute, returns fresh tensors, and steps aside (eager launches) whenever a replay could differ from the eager call — a forward hook
(viz/renderer.py:437), autograd, random noise — or go stale: weights edited in place are seen and the pass is captured again.
Call shapes are the reference drivers': gen_images.py:88-114 (one seed, three yaws, outputs appended to a list), gen_videos.py:129
(float64 `ws` from the spline), training/video_render.py / distributed_render.py (cached tri-planes read in place).  `pytest -m gpu`."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def G(gpu_device):
    from training import triplane
    torch.manual_seed(3)
    g = triplane.TriPlaneGenerator(triplane.tiny_spec()).eval().requires_grad_(False)
    return g.to(gpu_device)


@pytest.fixture(scope='module')
def G_full(gpu_device):
    from training import triplane
    torch.manual_seed(0)
    return triplane.TriPlaneGenerator().eval().requires_grad_(False).to(gpu_device)


def _ws(G, seeds, dev):
    from training import triplane
    z = torch.from_numpy(np.stack([np.random.RandomState(s).randn(G.z_dim) for s in seeds])).to(dev)
    return G.mapping(z, triplane.conditioning_label(dev).repeat(len(seeds), 1))


def _cams(yaws, dev):
    from training import triplane
    return torch.cat([triplane.camera_label(y, device=dev) for y in yaws])


def _launches():
    from torch_utils import hip_plugin
    return sum(hip_plugin.CALLS.values())


def test_second_call_captures_third_replays_and_both_equal_eager_bit_for_bit(G, gpu_device):
    from training import graph_cache
    graph_cache.reset(G.synthesis)
    ws = _ws(G, [0, 1], gpu_device)
    jit = torch.rand(2, G.synthesis.render_size ** 2, G.spec.num_steps, device=gpu_device)
    with graph_cache.disabled():
        want = [G.synthesis(ws, c=_cams([y, -y], gpu_device), noise_mode='const', return_seg=True, ray_jitter=jit) for y in (-0.5, 0.0, 0.5)]
    before = dict(graph_cache.STATS)
    got = []
    for y in (-0.5, 0.0, 0.5):          # gen_images.py:96-111: the outputs of all three calls are used after the loop
        c = _cams([y, -y], gpu_device)          # (the pose helpers are C-ABI launches of their own since ABI 8)
        n0 = _launches()
        got.append(G.synthesis(ws, c=c, noise_mode='const', return_seg=True, ray_jitter=jit))
        if y == 0.5:
            assert _launches() == n0, 'the third call with one signature must be a replay: no C-ABI launch'
    d = {k: graph_cache.STATS[k] - before.get(k, 0) for k in ('eager', 'capture', 'replay')}
    assert d == {'eager': 1, 'capture': 1, 'replay': 2}, d
    for (wi, wsg), (gi, gsg) in zip(want, got):
        assert torch.equal(wi, gi) and torch.equal(wsg, gsg)
    ptrs = {t.data_ptr() for pair in got for t in pair}
    assert len(ptrs) == 6, 'every call returns its own tensors (no aliasing of the static output buffers)'
    assert graph_cache.stats(G.synthesis)['graphs'] == 1


def test_jitter_drawn_per_call_from_the_same_generator_stream(G, gpu_device):
    """`ray_jitter=None`: eager draws torch.rand per call; the replay refills the static buffer from the same generator."""
    from training import graph_cache
    graph_cache.reset(G.synthesis)
    ws = _ws(G, [5], gpu_device); c = _cams([0.2], gpu_device)
    torch.manual_seed(11)
    with graph_cache.disabled():
        want = [G.synthesis(ws, c=c, return_seg=True) for _ in range(4)]
    torch.manual_seed(11)
    got = [G.synthesis(ws, c=c, return_seg=True) for _ in range(4)]
    assert graph_cache.stats(G.synthesis)['graphs'] == 1
    for k, ((wi, wsg), (gi, gsg)) in enumerate(zip(want, got)):
        assert torch.equal(wi, gi) and torch.equal(wsg, gsg), f'call {k}'
    assert not torch.equal(got[2][0], got[3][0]), 'fresh jitter per replay'


def test_forward_hook_anywhere_in_the_tree_keeps_the_call_eager(G, gpu_device):
    from training import graph_cache
    graph_cache.reset(G.synthesis)
    ws = _ws(G, [2], gpu_device); c = _cams([0.0], gpu_device)
    seen = []
    h = G.synthesis.b64.conv1.register_forward_hook(lambda m, i, o: seen.append(o.shape))      # viz/renderer.py:437
    try:
        for _ in range(3):
            n0 = _launches()
            G.synthesis(ws, c=c, ray_jitter=False)
            assert _launches() > n0
        assert len(seen) == 3 and graph_cache.stats(G.synthesis)['graphs'] == 0
    finally:
        h.remove()
    for _ in range(3):
        want = G.synthesis(ws, c=c, ray_jitter=False)
    assert graph_cache.stats(G.synthesis)['graphs'] == 1
    # a hook registered AFTER the pass was captured: the (optimistic) replay is thrown away, the eager pass runs — and calls the hook, once
    h = G.synthesis.vb8.register_forward_hook(lambda m, i, o: seen.append('vb8'))
    try:
        n0, d0 = _launches(), graph_cache.STATS['replay_discarded']
        got = G.synthesis(ws, c=c, ray_jitter=False)
        assert _launches() > n0 and seen[-1] == 'vb8' and seen.count('vb8') == 1 and graph_cache.STATS['replay_discarded'] == d0 + 1
        assert torch.equal(got, want)
    finally:
        h.remove()


def test_weights_edited_in_place_are_seen_and_the_pass_is_captured_again(G, gpu_device):
    from training import graph_cache
    graph_cache.reset(G.synthesis)
    ws = _ws(G, [7], gpu_device); c = _cams([0.1], gpu_device)
    for _ in range(3):
        a = G.synthesis(ws, c=c, ray_jitter=False, return_seg=True)
    assert graph_cache.stats(G.synthesis)['graphs'] == 1
    w = G.synthesis.b64.conv1.weight
    keep = w.detach().clone()
    try:
        with torch.no_grad():
            w.mul_(1.5)                                   # e.g. PTI fine-tuning between renders (inversion/scripts/run_pti.py:141-170)
        with graph_cache.disabled():
            want = G.synthesis(ws, c=c, ray_jitter=False, return_seg=True)
        assert not torch.equal(want[0], a[0])
        caps = graph_cache.STATS['capture']
        for k in range(3):
            got = G.synthesis(ws, c=c, ray_jitter=False, return_seg=True)
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), f'call {k} after the edit'
        assert graph_cache.STATS['capture'] == caps + 1 and graph_cache.stats(G.synthesis)['graphs'] == 1
    finally:
        with torch.no_grad():
            w.copy_(keep)


def test_autograd_and_random_noise_stay_eager(G, gpu_device):
    from training import graph_cache
    graph_cache.reset(G.synthesis)
    ws = _ws(G, [1], gpu_device).requires_grad_(True); c = _cams([0.0], gpu_device)
    for _ in range(3):
        img = G.synthesis(ws, c=c, ray_jitter=False)
    assert img.requires_grad and graph_cache.stats(G.synthesis)['graphs'] == 0
    ws = ws.detach()
    for _ in range(3):
        G.synthesis(ws, c=c, ray_jitter=False, noise_mode='random')
    assert graph_cache.stats(G.synthesis)['graphs'] == 0
    G.synthesis.auto_graph = False
    try:
        for _ in range(3):
            G.synthesis(ws, c=c, ray_jitter=False)
        assert graph_cache.stats(G.synthesis)['graphs'] == 0
    finally:
        del G.synthesis.auto_graph


def test_float64_ws_and_return_forms(G, gpu_device):
    """gen_videos.py:127-129 hands float64 `ws` (scipy spline); return_dict / return_raw carry views of one tensor."""
    from training import graph_cache
    graph_cache.reset(G.synthesis)
    ws = _ws(G, [3], gpu_device).double(); c = _cams([-0.3], gpu_device)
    with graph_cache.disabled():
        want = G.synthesis(ws=ws, c=c, noise_mode='const', return_dict=True, ray_jitter=False)
    for _ in range(3):
        got = G.synthesis(ws=ws, c=c, noise_mode='const', return_dict=True, ray_jitter=False)
    assert graph_cache.stats(G.synthesis)['graphs'] == 1
    for k in ('image', 'image_seg', 'image_raw', 'image_depth'):
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape and torch.equal(got[k], want[k]), k
    for a, b in zip(got['planes'], want['planes']):
        assert torch.equal(a, b) and a.stride() == b.stride()


def test_cached_planes_are_read_in_place_and_planes_call_is_replayed(G, gpu_device):
    """training/distributed_render.py, training/video_render.py: tri-planes computed once per seed, then many poses."""
    from training import graph_cache
    graph_cache.reset(G.synthesis)
    ws = _ws(G, [4, 9], gpu_device)
    with graph_cache.disabled():
        voxel_ws, _ = G.synthesis.split_ws(ws)
        want_planes = G.synthesis.backbone(voxel_ws, noise_mode='const')
    for _ in range(3):
        planes = G.synthesis.planes(ws)
    assert graph_cache.stats(G.synthesis)['graphs'] == 1
    for a, b in zip(planes, want_planes):
        assert torch.equal(a, b) and a.stride() == b.stride()
    buf = tuple(p.clone() for p in planes)
    for yaw in (-0.4, 0.0, 0.4, 0.2):
        c = _cams([yaw, -yaw], gpu_device)
        with graph_cache.disabled():
            want = G.synthesis(ws, c=c, cached_planes=buf, ray_jitter=False, return_seg=True)
        got = G.synthesis(ws, c=c, cached_planes=buf, ray_jitter=False, return_seg=True)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert graph_cache.stats(G.synthesis)['graphs'] == 2
    # new contents at the same addresses (the next seed's tri-planes written into the driver's buffers) are what the replay reads
    ws2 = _ws(G, [12, 13], gpu_device)
    for b, p in zip(buf, G.synthesis.planes(ws2)):
        b.copy_(p)
    c = _cams([0.3, 0.1], gpu_device)
    with graph_cache.disabled():
        want = G.synthesis(ws2, c=c, cached_planes=buf, ray_jitter=False, return_seg=True)
    n0 = _launches()
    got = G.synthesis(ws2, c=c, cached_planes=buf, ray_jitter=False, return_seg=True)
    assert _launches() == n0
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_lru_bound_and_conv_arithmetic_in_the_signature(G, gpu_device, monkeypatch):
    from torch_utils import hip_plugin
    from training import graph_cache
    graph_cache.reset(G.synthesis)
    monkeypatch.setenv('IDE3D_AUTO_GRAPH_MAX', '2')
    c = _cams([0.0], gpu_device)
    for n in (1, 2, 3):
        ws = _ws(G, list(range(n)), gpu_device)
        for _ in range(2):
            G.synthesis(ws, c=c.repeat(n, 1), ray_jitter=False)
    assert graph_cache.stats(G.synthesis)['graphs'] == 2
    keep = hip_plugin.conv_arithmetic()
    try:
        hip_plugin.conv_arithmetic('fp32')
        ws = _ws(G, [0, 1, 2], gpu_device)
        caps = graph_cache.STATS['capture']
        for _ in range(3):          # one eviction has happened above: a new signature now needs two eager sightings (capture back-off)
            G.synthesis(ws, c=c.repeat(3, 1), ray_jitter=False)
        assert graph_cache.STATS['capture'] == caps + 1, 'another arithmetic is another launch sequence'
    finally:
        hip_plugin.conv_arithmetic(keep)


def test_reference_gen_images_loop_full_size_replay_equals_eager(G_full, gpu_device):
    """The loop of gen_images.py:88-114 at full size in the library-default arithmetic: mapping per seed, three yaws, `render_params`
    as the driver passes them; the replayed images are bit-equal to eager launches of the same calls."""
    import math
    from training import graph_cache
    G = G_full
    graph_cache.reset(G.synthesis)
    for seed in (0, 1):
        torch.manual_seed(seed)
        ws = _ws(G, [seed], gpu_device)
        imgs, segs = [], []
        for yaw in (-0.5, 0.0, 0.5):
            rp = {'h_mean': yaw + math.pi * 0.5, 'v_mean': math.pi * 0.5, 'h_stddev': 0., 'v_stddev': 0., 'fov': 18, 'num_steps': 96}
            img, seg = G.synthesis(ws, c=_cams([yaw], gpu_device), render_params=rp, noise_mode='const', return_seg=True)
            imgs.append(img); segs.append(seg)
        torch.manual_seed(seed)
        with graph_cache.disabled():
            for k, yaw in enumerate((-0.5, 0.0, 0.5)):
                rp = {'h_mean': yaw + math.pi * 0.5, 'v_mean': math.pi * 0.5, 'h_stddev': 0., 'v_stddev': 0., 'fov': 18, 'num_steps': 96}
                img, seg = G.synthesis(ws, c=_cams([yaw], gpu_device), render_params=rp, noise_mode='const', return_seg=True)
                assert torch.equal(img, imgs[k]) and torch.equal(seg, segs[k]), (seed, yaw)
    assert graph_cache.stats(G.synthesis)['graphs'] == 1


def test_two_threads_on_their_own_streams_share_the_module(G, gpu_device):
    """Two host threads render with ONE generator on two streams (the eager path supports it: per-thread prefetch tables, per-stream
    workspaces).  The stream is part of the signature, so each thread captures and replays its own graph with its own static buffers and
    memory pool; the module's cache is entered by one thread at a time."""
    import threading
    from training import graph_cache
    graph_cache.reset(G.synthesis)
    c = _cams([0.1], gpu_device)
    ws = [_ws(G, [20 + k], gpu_device) for k in range(2)]
    with graph_cache.disabled():
        want = [G.synthesis(w, c=c, ray_jitter=False, return_seg=True) for w in ws]
    torch.cuda.synchronize()
    got, errors = [None, None], []

    def worker(k):
        try:
            st = torch.cuda.Stream(device=gpu_device)
            with torch.cuda.stream(st), torch.no_grad():
                for _ in range(6):
                    out = G.synthesis(ws[k], c=c, ray_jitter=False, return_seg=True)
                st.synchronize()
            got[k] = out
        except Exception as e:          # noqa: BLE001 - reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(2):
        assert torch.equal(got[k][0], want[k][0]) and torch.equal(got[k][1], want[k][1]), f'thread {k}'
    assert graph_cache.stats(G.synthesis)['graphs'] == 2


def test_replays_do_not_grow_memory_and_dropped_graphs_release_theirs(G, gpu_device, monkeypatch):
    """300 calls over four alternating signatures (LRU bound 3: one is evicted and captured again — ever more rarely: every eviction doubles
    the eager sightings the next capture needs, so a rotating caller does not pay a capture per rotation) leave the allocator where 60 calls
    left it."""
    from training import graph_cache
    graph_cache.reset(G.synthesis)
    monkeypatch.setenv('IDE3D_AUTO_GRAPH_MAX', '3')
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated(gpu_device)
    c = _cams([0.0], gpu_device)
    wss = {n: _ws(G, list(range(n)), gpu_device) for n in (1, 2, 3, 4)}

    def burst(k):
        for i in range(k):
            n = 1 + i % 4
            out = G.synthesis(wss[n], c=c.repeat(n, 1), ray_jitter=False, return_seg=True)
        del out
        torch.cuda.synchronize()
        return torch.cuda.memory_allocated(gpu_device)

    caps = graph_cache.STATS['capture']
    m60 = burst(60)
    m300 = burst(240)
    assert graph_cache.stats(G.synthesis)['graphs'] == 3
    assert graph_cache.STATS['capture'] - caps <= 12, 'capture thrash: 300 calls over 4 signatures must not capture 60 times'

    assert m300 <= m60 + (1 << 20), f'allocator grew by {(m300 - m60) / 2**20:.1f} MiB over 240 more calls'
    graph_cache.reset(G.synthesis)
    import gc; gc.collect()
    torch.cuda.synchronize()
    assert graph_cache.stats(G.synthesis)['graphs'] == 0 and torch.cuda.memory_allocated(gpu_device) <= m300


def test_more_captures_than_torch_has_pooled_streams(G, gpu_device, monkeypatch):
    """Regression (round 5): `torch.cuda.Stream()` hands out 32 pooled handles round-robin, so the 33rd capture of a process used to be
    captured on a handle that was already the style side stream (or an older graph's capture stream): a malformed capture and a host-side
    segfault in hipGraphLaunch.  Captures and the side stream now use streams of the library's own (`hip_plugin.private_stream`): 40
    capture / evict cycles beside a graph that stays alive, every replay still equal to the eager pass."""
    from torch_utils import hip_plugin
    from training import graph_cache
    graph_cache.reset(G.synthesis)
    monkeypatch.setenv('IDE3D_AUTO_GRAPH_MAX', '2')
    c = _cams([0.0], gpu_device)
    keep_ws = _ws(G, [1, 2, 3, 4, 5], gpu_device)
    with graph_cache.disabled():
        keep_want = G.synthesis(keep_ws, c=c.repeat(5, 1), ray_jitter=False)
    wss = {n: _ws(G, list(range(n)), gpu_device) for n in (1, 2, 3)}
    with graph_cache.disabled():
        want = {n: G.synthesis(wss[n], c=c.repeat(n, 1), ray_jitter=False) for n in (1, 2, 3)}
    caps = graph_cache.STATS['capture']
    i = 0
    while graph_cache.STATS['capture'] - caps < 40:
        n = 1 + i % 3; i += 1
        graph_cache._caches[G.synthesis].evictions = 0          # (switch the anti-thrash back-off off: this test WANTS a capture per rotation)
        assert torch.equal(G.synthesis(wss[n], c=c.repeat(n, 1), ray_jitter=False), want[n])
        if i % 3 == 0:          # the five-image signature is used often enough to stay in the LRU pair: its graph outlives 40 captures
            assert torch.equal(G.synthesis(keep_ws, c=c.repeat(5, 1), ray_jitter=False), keep_want)
    assert i < 400
    s1, s2 = hip_plugin.private_stream(gpu_device, 'graph capture'), hip_plugin.private_stream(gpu_device, 'style prefetch')
    pooled = {torch.cuda.Stream(device=gpu_device).cuda_stream for _ in range(40)}
    assert s1.cuda_stream != s2.cuda_stream and s1.cuda_stream not in pooled and s2.cuda_stream not in pooled
