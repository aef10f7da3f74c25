"""hipGraph capture-and-replay of `G.synthesis` on the caller's behalf.

The reference's drivers call `G.synthesis(ws, c=c, ...)` once per image in a plain Python loop (gen_images.py:88-114,
gen_videos.py:114-139).  On an MI355X one such call is ~80 launches of 5-300 us of GPU work; enqueued one by one through
ctypes the HOST bounds the loop (2.5 ms per batch-1 image for 1.75 ms of GPU time).  A caller who builds a
`triplane.GraphedRenderer` gets the replayed rate, but an unchanged driver cannot.  This module gives it to every caller:

    call 1 with a signature   -> eager launches (a one-off call never pays for a capture)
    call 2 with the same one  -> warm-up pass inside this graph's workspace scope, capture, replay
    call 3 ...                -> copy ws / c (/ jitter) into the graph's static inputs, replay, return fresh copies of the outputs

Signature = everything the captured launch sequence depends on: batch size, flags, render parameters, the arithmetic of the
convolutions, grad mode, the current stream, the addresses of cached tri-planes (read in place: 50 MB per image), and a stamp of
every parameter and buffer of the module tree (`_version` and storage address: an in-place edit or a `load_state_dict` makes a
new signature and drops the stale graphs).  Calls that cannot be replayed faithfully run eagerly, as before:

    * a forward (pre-)hook on any module of the tree (viz/renderer.py:437 registers one per module to read activations) or a global one;
    * autograd in play (grad mode on and `ws` or a parameter requires grad);
    * `noise_mode='random'`, `nerf_noise`, the hierarchical pass (fresh device draws inside the pass);
    * a capture already in progress on this stream (`GraphedRenderer`), CPU tensors, `IDE3D_AUTO_GRAPH=0`, `module.auto_graph = False`,
      or inside `with graph_cache.disabled():`.

Outputs are COPIES of the graph's static output buffers (gen_images.py:110 appends the images of three calls to a list before it
uses them), so the semantics are those of the eager call.  Stratified jitter: the eager pass draws `torch.rand` on the device per
call; the replay fills the static jitter buffer with `uniform_()` from the same generator — the same stream of draws.

What `_version` cannot see (writes through `.data`, through numpy views, by foreign kernels) is invisible here exactly as it is to the
eager path's packed-weight caches (`networks._stamp`): call `graph_cache.reset(module)` after such an edit.
"""

import collections
import contextlib
import os
import threading
import warnings
import weakref

import torch

_tls = threading.local()
_caches = weakref.WeakKeyDictionary()        # synthesis module -> _ModuleGraphs (never pickled / deep-copied with the generator)
_pools = {}                                  # (device index, stream handle) -> graph memory pool shared by the graphs replayed on that stream

STATS = collections.Counter()                # 'eager', 'capture', 'replay', 'ineligible:<why>' — read by tests and bench.py


_PASS_ENV = ('IDE3D_NO_SKIP_MERGE', 'IDE3D_NO_STYLE_PREFETCH', 'IDE3D_NO_STYLE_BATCH', 'IDE3D_GATHER_PC', 'IDE3D_NO_LOWRES_GROUP', 'IDE3D_LOWRES_PERSISTENT')


def _env_int(name, default):
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


def enabled():
    return os.environ.get('IDE3D_AUTO_GRAPH', '1') != '0' and not getattr(_tls, 'off', 0)


@contextlib.contextmanager
def disabled():
    """`with graph_cache.disabled():` — every `G.synthesis` call inside launches eagerly (tests that want the eager path; warm-up
    passes of `GraphedRenderer`)."""
    _tls.off = getattr(_tls, 'off', 0) + 1
    try:
        yield
    finally:
        _tls.off -= 1


def reset(module=None):
    """Drop the captured graphs of `module` (all modules when None)."""
    if module is None:
        for c in list(_caches.values()):
            c.clear()
    elif module in _caches:
        _caches[module].clear()


def stats(module):
    c = _caches.get(module)
    return {'graphs': 0 if c is None else len(c.entries), 'seen': 0 if c is None else len(c.seen)}


def tree_stamp(root):
    """(hash of every parameter's / buffer's (version, address), any requires_grad, any forward hook) of the module tree, one pass.
    ~60 us for the full generator: cheap next to the 1.4 ms of GPU time of a batch-1 pass."""
    acc = []
    push = acc.append
    requires_grad = False
    hooked = False
    stack = [root]
    while stack:
        m = stack.pop()
        if m._forward_hooks or m._forward_pre_hooks:
            hooked = True
        for p in m._parameters.values():
            if p is not None:
                push(p._version); push(p.data_ptr())
                if p.requires_grad:
                    requires_grad = True
        for b in m._buffers.values():
            if b is not None:
                push(b._version); push(b.data_ptr())
        for sub in m._modules.values():
            if sub is not None:
                stack.append(sub)
    return hash(tuple(acc)), requires_grad, hooked


def _global_hooks():
    mod = torch.nn.modules.module
    return bool(getattr(mod, '_global_forward_hooks', None)) or bool(getattr(mod, '_global_forward_pre_hooks', None))


class _Captured:
    """One captured pass: static inputs, the graph, its static outputs.  Also the owner of the pass's workspaces
    (`hip_plugin.workspace_scope`): packed weights and split-K partials of this graph are its own."""
    __slots__ = ('graph', 'ws', 'c', 'jitter', 'jitter_given', 'out', 'planes', 'tensors', 'aliases', 'versions', '__weakref__')

    def untouched(self):
        """No parameter / buffer the pass was captured with has been edited in place since (~15 us).  Their storages cannot have been
        freed either: `aliases` share them.  What this cannot see — replaced Parameter objects, `.data` swaps, hooks — the full walk finds."""
        v = 0
        for t in self.tensors:
            v += t._version
        return v == self.versions


class _ModuleGraphs:
    def __init__(self):
        self.entries = collections.OrderedDict()     # signature -> _Captured (LRU order)
        self.seen = {}                               # signature -> eager sightings so far
        self.stamp = None
        self.evictions = 0                           # LRU evictions so far: every one doubles the sightings a new signature needs (no capture thrash)
        self.lock = threading.RLock()                # one thread at a time per module: a replay and the copy-out of its static outputs are one step

    def clear(self):
        if self.entries:          # replays may still be running (and reading the graphs' workspaces): let them finish before anything is freed
            torch.cuda.synchronize(next(iter(self.entries.values())).ws.device)
        self.entries.clear()
        self.seen.clear()
        self.evictions = 0


def _out_tensors(out):
    if isinstance(out, dict):
        return list(out.values())
    return list(out) if isinstance(out, (tuple, list)) else [out]


def _fresh(out, caller_planes):
    """Copies of the static outputs with the eager call's structure.  Views of one base tensor (the dual head writes image and
    segmentation into channel ranges of one tensor) are copied once and re-sliced."""
    memo = {}

    def cp(t):
        if not torch.is_tensor(t):
            return t
        base = t._base
        if (base is not None and base.ndim >= 2 and base.is_contiguous() and t.ndim == base.ndim and t.stride() == base.stride()
                and t.shape[0] == base.shape[0] and t.shape[2:] == base.shape[2:]):
            nb = memo.get(id(base))
            if nb is None:
                nb = memo[id(base)] = base.clone()
            off = (t.data_ptr() - base.data_ptr()) // (base.stride(1) * base.element_size())
            return nb[:, off:off + t.shape[1]]
        return t.clone()

    if isinstance(out, dict):
        res = {}
        for k, v in out.items():
            if k == 'planes':
                res[k] = caller_planes if caller_planes is not None else tuple(cp(p) for p in v)
            else:
                res[k] = cp(v)
        return res
    if isinstance(out, (tuple, list)):
        return type(out)(cp(t) for t in out)
    return cp(out)


def run(module, impl, ws, c, render_params, noise_mode, flags, force_fp32, ray_jitter, cached_planes, extra_kwargs, kind='synthesis'):
    """Called by `TriplaneSynthesisNetwork.forward` (kind 'synthesis') and `.planes` (kind 'backbone': ws -> the two tri-planes, no
    camera, no jitter).  `impl(ws, c, ray_jitter, cached_planes)` is the eager pass with every other argument bound.  Returns the
    outputs (replayed or eager)."""
    why = _ineligible(module, ws, c, render_params, noise_mode, ray_jitter, cached_planes, extra_kwargs)
    if why is not None:
        STATS['ineligible:' + why] += 1
        return impl(ws, c, ray_jitter, cached_planes)

    from torch_utils import hip_plugin
    from training import networks
    cache = _caches.get(module)
    if cache is None:
        cache = _caches.setdefault(module, _ModuleGraphs())
    with cache.lock:
        return _run_locked(cache, module, impl, ws, c, render_params, noise_mode, flags, force_fp32, ray_jitter, cached_planes, kind)


def _run_locked(cache, module, impl, ws, c, render_params, noise_mode, flags, force_fp32, ray_jitter, cached_planes, kind):
    from torch_utils import hip_plugin
    from training import networks
    sp = module.spec
    n = ws.shape[0]
    steps = render_params.get('num_steps') or sp.num_steps
    jit_kind = 'none' if ray_jitter is False else ('draw' if ray_jitter is None else 'given')
    planes_key = None
    if cached_planes is not None:
        planes_key = tuple((p.data_ptr(), tuple(p.shape), p.stride()) for p in cached_planes)
    dev = ws.device
    sig = (kind, n, tuple(ws.shape[1:]), None if c is None else (tuple(c.shape), c.dtype), noise_mode, flags, bool(force_fp32), jit_kind, planes_key,
           render_params.get('fov'), steps, render_params.get('ray_start'), render_params.get('ray_end'),
           hip_plugin.conv_arithmetic(), networks.use_hip_modconv, torch.is_grad_enabled(), dev.index, hip_plugin._stream_handle(dev),
           bool(getattr(module, 'style_prefetch', True)),
           # what else selects the launch path per call: train / eval of the tree's blocks (`fused_modconv = not self.training`,
           # networks.py) and the environment switches the pass reads while it runs
           module.training, tuple(ch.training for ch in module.children()), tuple(os.environ.get(k) for k in _PASS_ENV))

    def current(stamp, requires_grad, hooked):
        """None when a replay is what the eager call would compute, else the reason it is not."""
        if hooked or _global_hooks():
            return 'forward hook'
        if torch.is_grad_enabled() and (requires_grad or ws.requires_grad):
            return 'autograd'
        if cache.stamp != stamp:
            return 'parameters changed'
        return None

    ent = cache.entries.get(sig)
    # cheap rejections BEFORE the optimistic replay: with autograd the static input would join the caller's graph, and a hook on the root
    # (the common place) needs no tree walk to be seen
    replayable = not (torch.is_grad_enabled() and ws.requires_grad) and not module._forward_hooks and not module._forward_pre_hooks and not _global_hooks()
    if ent is not None and replayable and ent.untouched():
        # Optimistic replay: the launch goes out first, the ~0.1 ms walk over the module tree (hooks, requires_grad, every parameter's
        # version and address) runs while the GPU works, and the copies are handed out only if the walk finds nothing.  In the drivers'
        # loops the host is on the critical path between a blocking `z.to(device)` and the first launch of the pass (gen_images.py:91-109).
        cache.entries.move_to_end(sig)
        gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
        offset = gen.get_offset() if (ent.jitter is not None and not ent.jitter_given) else None
        out = _replay(ent, ws, c, ray_jitter, cached_planes)
        why = current(*tree_stamp(module))
        if why is None:
            STATS['replay'] += 1
            return out
        del out                      # a pure function of the static inputs was evaluated for nothing; the eager pass below is the answer
        if offset is not None:
            gen.set_offset(offset)   # the discarded replay's jitter draw is handed back: the eager pass below draws what a purely eager caller would
        STATS['replay_discarded'] += 1

    stamp, requires_grad, hooked = tree_stamp(module)
    if cache.stamp != stamp:           # a parameter / buffer changed (or moved): every captured pass is stale for good
        cache.clear()
        cache.stamp = stamp
    why = current(stamp, requires_grad, hooked)
    if why is not None:
        STATS['ineligible:' + why] += 1
        return impl(ws, c, ray_jitter, cached_planes)

    # A capture costs ~3 passes + a device synchronisation.  A caller that rotates through more signatures than the LRU holds would pay it
    # over and over: every eviction doubles the number of eager sightings the next new signature needs (1, 2, 4, ... 64).
    seen = cache.seen.get(sig, 0)
    if seen < _env_int('IDE3D_AUTO_GRAPH_AFTER', 1) << min(cache.evictions, 6):
        if len(cache.seen) > 256:
            cache.seen.clear()
        cache.seen[sig] = seen + 1
        STATS['eager'] += 1
        return impl(ws, c, ray_jitter, cached_planes)
    try:
        with hip_plugin.capture_lock:
            ent = _capture(module, impl, ws, c, ray_jitter, cached_planes, steps, sig)
    except Exception as e:      # never a silent slow path: say so once per signature, then stay eager for it
        warnings.warn(f'ide3d graph_cache: capture of G.synthesis failed ({type(e).__name__}: {e}); this call signature stays eager')
        cache.seen[sig] = -(1 << 60)
        STATS['capture_failed'] += 1
        return impl(ws, c, ray_jitter, cached_planes)
    cache.entries[sig] = ent
    cache.seen.pop(sig, None)
    while len(cache.entries) > max(1, _env_int('IDE3D_AUTO_GRAPH_MAX', 6)):
        torch.cuda.synchronize(ws.device)              # replays of the graph that is about to go may still be running; so may readers of its workspaces
        cache.entries.popitem(last=False)
        cache.evictions += 1
    STATS['capture'] += 1
    STATS['replay'] += 1
    return _replay(ent, ws, c, ray_jitter, cached_planes)


def _replay(ent, ws, c, ray_jitter, cached_planes):
    with torch.no_grad():            # the static inputs never join an autograd graph
        ent.ws.copy_(ws, non_blocking=True)
        if ent.c is not None:
            ent.c.copy_(c, non_blocking=True)
        if ent.jitter is not None:
            if ent.jitter_given:
                ent.jitter.copy_(ray_jitter, non_blocking=True)
            else:
                ent.jitter.uniform_()
    ent.graph.replay()
    return _fresh(ent.out, cached_planes)


def _ineligible(module, ws, c, render_params, noise_mode, ray_jitter, cached_planes, extra_kwargs):
    if not enabled() or not getattr(module, 'auto_graph', True):
        return 'switched off'
    if not (torch.is_tensor(ws) and ws.is_cuda and ws.ndim == 3 and (c is None or (torch.is_tensor(c) and c.device == ws.device and c.ndim == 2))):
        return 'not device tensors'
    if ws.device.index is not None and ws.device.index != torch.cuda.current_device():
        return 'tensors on a device that is not the current one'          # the eager launches guard the device per call; a replay would not
    if torch.cuda.is_current_stream_capturing():
        return 'capture in progress'
    if noise_mode not in ('const', 'none'):
        return 'random noise'
    if extra_kwargs:
        return 'unknown keyword arguments'
    sp = module.spec
    if render_params.get('nerf_noise') or render_params.get('importance_u') is not None:
        return 'device draws inside the pass'
    hier = render_params.get('hierarchical')
    if hier or (hier is None and sp.hierarchical):
        return 'hierarchical pass'
    for k in ('fov', 'num_steps', 'ray_start', 'ray_end'):
        v = render_params.get(k)
        if v is not None and not isinstance(v, (int, float)):
            return 'non-scalar render parameter'
    if ray_jitter is not None and ray_jitter is not False:
        steps = render_params.get('num_steps') or sp.num_steps
        if not (torch.is_tensor(ray_jitter) and ray_jitter.device == ws.device
                and tuple(ray_jitter.shape) == (ws.shape[0], sp.render_size ** 2, steps)):
            return 'jitter shape'
    if cached_planes is not None:
        if not (isinstance(cached_planes, (tuple, list)) and len(cached_planes) == 2
                and all(torch.is_tensor(p) and p.device == ws.device and p.dtype == torch.float32 and not p.requires_grad for p in cached_planes)):
            return 'cached planes'
    return None


def _capture(module, impl, ws, c, ray_jitter, cached_planes, steps, sig):
    from torch_utils import hip_plugin
    dev = ws.device
    sp = module.spec
    ent = _Captured()
    ent.ws = ws.detach().to(torch.float32).contiguous().clone()
    ent.c = None if c is None else c.detach().clone()
    ent.planes = cached_planes            # keeps the tensors the graph reads in place alive for as long as the graph
    ent.jitter_given = ray_jitter is not None and ray_jitter is not False
    if ray_jitter is False:
        ent.jitter = None
    else:
        ent.jitter = torch.empty([ws.shape[0], sp.render_size ** 2, steps], dtype=torch.float32, device=dev)
        if ent.jitter_given:
            ent.jitter.copy_(ray_jitter)
        else:
            ent.jitter.fill_(0.5)         # the warm-up's values are irrelevant; no draw is taken from the caller's generator
    jit_arg = False if ent.jitter is None else ent.jitter
    main = torch.cuda.current_stream(dev)
    stream = hip_plugin.private_stream(dev, 'graph capture')          # never one of torch's pooled handles (hip_plugin.private_stream)
    stream.wait_stream(main)
    scope = hip_plugin.workspace_scope(ent)
    with torch.cuda.stream(stream), torch.no_grad(), scope, disabled():
        impl(ent.ws, ent.c, jit_arg, cached_planes)          # packs weights / sizes workspaces inside this graph's scope
    main.wait_stream(stream)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    pool = None
    if os.environ.get('IDE3D_AUTO_GRAPH_SHARED_POOL', '0') == '1':
        pool_key = (dev.index, hip_plugin._stream_handle(dev))
        pool = _pools.get(pool_key)
        if pool is None:
            pool = _pools[pool_key] = torch.cuda.graph_pool_handle()
    # thread_local: another host thread of the application (a data loader, a second renderer) may keep calling the runtime while this one captures
    with torch.no_grad(), scope, disabled(), torch.cuda.graph(graph, stream=stream, pool=pool, capture_error_mode='thread_local'):
        ent.out = impl(ent.ws, ent.c, jit_arg, cached_planes)
    ent.graph = graph
    # the memory the graph reads through raw pointers stays allocated for as long as the graph: the tensors themselves and aliases of their
    # storages (a later `p.data = other` re-seats the Parameter object, not the alias)
    ent.tensors = [t for m in module.modules() for t in list(m._parameters.values()) + list(m._buffers.values()) if t is not None]
    ent.aliases = [t.detach() for t in ent.tensors]
    ent.versions = sum(t._version for t in ent.tensors)
    return ent
