"""Host logic of training/graph_cache.py that needs no GPU: the module-tree stamp (what makes a captured pass stale), eligibility, the
copy-out of static outputs, and that CPU calls are untouched.  The replays themselves: tests/test_gpu_graph_cache.py (`-m gpu`)."""

import torch

from training import graph_cache, triplane


def _tiny():
    torch.manual_seed(0)
    return triplane.TriPlaneGenerator(triplane.tiny_spec()).eval()


def test_tree_stamp_sees_in_place_edits_moves_hooks_and_requires_grad():
    G = _tiny()
    s0, rg, hooked = graph_cache.tree_stamp(G.synthesis)
    assert rg and not hooked                       # a freshly built generator's parameters require grad
    assert graph_cache.tree_stamp(G.synthesis)[0] == s0
    with torch.no_grad():
        G.synthesis.b64.conv1.weight.mul_(2.0)     # in-place edit: `_version` moves
    s1 = graph_cache.tree_stamp(G.synthesis)[0]
    assert s1 != s0
    G.synthesis.b64.conv1.noise_const.add_(1.0)    # buffers count too (const noise is an input of the pass)
    s2 = graph_cache.tree_stamp(G.synthesis)[0]
    assert s2 != s1
    G.synthesis.b64.conv1.weight = torch.nn.Parameter(G.synthesis.b64.conv1.weight.detach().clone())      # replaced object: another address
    assert graph_cache.tree_stamp(G.synthesis)[0] != s2
    G.requires_grad_(False)
    assert graph_cache.tree_stamp(G.synthesis)[1] is False
    h = G.synthesis.vb8.torgb.register_forward_hook(lambda m, i, o: None)      # viz/renderer.py:437 hooks every module
    assert graph_cache.tree_stamp(G.synthesis)[2] is True
    h.remove()
    assert graph_cache.tree_stamp(G.synthesis)[2] is False
    h = G.synthesis.renderer.register_forward_pre_hook(lambda m, i: None)
    assert graph_cache.tree_stamp(G.synthesis)[2] is True
    h.remove()


def test_cpu_calls_never_reach_the_cache_and_results_are_unchanged():
    G = _tiny().requires_grad_(False)
    ws = torch.randn(2, G.num_ws, G.w_dim)
    c = torch.cat([triplane.camera_label(-0.3), triplane.camera_label(0.3)])
    before = dict(graph_cache.STATS)
    a = [G.synthesis(ws, c=c, ray_jitter=False, return_seg=True) for _ in range(3)]
    assert dict(graph_cache.STATS) == before and graph_cache.stats(G.synthesis) == {'graphs': 0, 'seen': 0}
    assert torch.equal(a[0][0], a[2][0]) and torch.equal(a[0][1], a[2][1])
    planes = G.synthesis.planes(ws)
    voxel_ws, _ = G.synthesis.split_ws(ws)
    want = G.synthesis.backbone(voxel_ws, noise_mode='const')
    assert all(torch.equal(p, q) for p, q in zip(planes, want))
    got = G.synthesis(ws, c=c, ray_jitter=False, return_seg=True, cached_planes=planes)
    assert torch.equal(got[0], a[0][0]) and torch.equal(got[1], a[0][1])


def test_ineligible_reasons():
    G = _tiny().requires_grad_(False)
    m = G.synthesis
    ws = torch.randn(1, G.num_ws, G.w_dim); c = triplane.camera_label(0.0)
    why = lambda **kw: graph_cache._ineligible(m, kw.get('ws', ws), kw.get('c', c), kw.get('rp', {}), kw.get('noise_mode', 'const'),
                                               kw.get('jitter'), kw.get('planes'), kw.get('extra', {}))
    assert why() == 'not device tensors'           # CPU tensors (the only kind this container has)
    with graph_cache.disabled():
        assert why() == 'switched off'
        with graph_cache.disabled():
            assert not graph_cache.enabled()
        assert not graph_cache.enabled()
    assert graph_cache.enabled()
    m.auto_graph = False
    assert why() == 'switched off'
    del m.auto_graph


def test_fresh_copies_keep_the_structure_of_the_static_outputs():
    both = torch.arange(2 * 8 * 4 * 4, dtype=torch.float32).reshape(2, 8, 4, 4).clone()       # a tensor of its own, like the dual head's output
    img, seg = both[:, :3], both[:, 3:]
    depth = torch.ones(2, 1, 4, 4)
    cl = torch.randn(2, 6, 4, 4).contiguous(memory_format=torch.channels_last)
    out = graph_cache._fresh(dict(image=img, image_seg=seg, image_depth=depth, planes=(cl, cl)), None)
    assert torch.equal(out['image'], img) and torch.equal(out['image_seg'], seg) and out['image'].data_ptr() != img.data_ptr()
    assert out['image']._base is out['image_seg']._base and out['image'].stride() == img.stride()      # one copy of the shared base, re-sliced
    assert out['planes'][0].stride() == cl.stride() and out['planes'][0].data_ptr() != cl.data_ptr()
    mine = (torch.zeros(1), torch.zeros(1))
    assert graph_cache._fresh(dict(image=img, planes=(cl, cl)), mine)['planes'] is mine               # the caller's cached planes go back as they came
    a, b = graph_cache._fresh((img, seg), None)
    assert torch.equal(a, img) and torch.equal(b, seg) and a._base is b._base
    t = graph_cache._fresh(depth, None)
    assert torch.equal(t, depth) and t.data_ptr() != depth.data_ptr()
