"""StyleGAN2 building blocks of the IDE-3D tri-plane generator.

The released IDE-3D generator lives only inside its pickle; the blocks it is made of
correspond to the StyleNeRF-derived `inversion/networks.py` of the reference: `modulated_conv2d` (:55),
`FullyConnectedLayer` (:136), `Conv2dLayer` (:170), `MappingNetwork` (:246), `SynthesisLayer` (:330),
`ToRGBLayer` (:670) and the dual-path `SegSynthesisBlock` (:966).  This module re-states those blocks
(2-D mode, default up-sampling, 'skip' / 'orig' architectures) with identical parameter and buffer
names, so state dicts are interchangeable with the reference's blocks, and `training.networks.*` is
the module path the reference's blocks look up by default (`layer_name`, networks.py:765,1016).

MI355X specifics: on device tensors in inference (no autograd graph) a stride-1 modulated convolution
and its epilogue (noise, bias, leaky ReLU, gain, clamp) are ONE fp32-MFMA implicit-GEMM launch
(`csrc/modconv.hip`); up-sampling layers run the reference's transposed-convolution + FIR strategy
with the FIR, the demodulation/noise and the bias-activation on HIP kernels.  With autograd the blocks
evaluate the same mathematics through differentiable ops.
"""

import math
import weakref

import numpy as np
import torch

from dnnlib import util
from torch_utils import custom_ops
from torch_utils import misc
from torch_utils import persistence
from torch_utils.ops import bias_act
from torch_utils.ops import conv2d_resample
from torch_utils.ops import fma
from torch_utils.ops import upfirdn2d

_modconv_plugin = None

# Route stride-1 modulated convolutions on device tensors through csrc/modconv.hip (inference only).
use_hip_modconv = True


def _upfirdn_plugin():
    upfirdn2d._init()
    return upfirdn2d._plugin


_style_plugin = None
_resample_plugin = None
_mapping_plugin = None


def _resample_init():
    global _resample_plugin
    if _resample_plugin is None:
        _resample_plugin = custom_ops.get_plugin(module_name='resample_plugin', sources=['resample.hip'])
    return True


def _style_init():
    global _style_plugin
    if _style_plugin is None:
        _style_plugin = custom_ops.get_plugin(module_name='style_plugin', sources=['style.hip'])
    return True


def _stamp(t):
    """What has to be unchanged for a tensor derived from parameter `t` to be still valid: `Module.to(device)` and
    `param.data = ...` keep the Parameter object and its `_version` but change the storage, so device and address count."""
    return (t._version, t.device, t.data_ptr())


def _evict_dead(cache, limit=512):
    """Bound a derived-tensor cache by dropping entries whose source tensor is gone.  Live entries are never freed: a
    captured hipGraph holds raw pointers into them (a blanket `.clear()` would leave its replays reading freed memory)."""
    if len(cache) > limit:
        for k in [k for k, e in cache.items() if e[0]() is None]:
            del cache[k]


def _wsq_t(weight):
    """[Cin, Cout] = sum_k W[o, i, k]^2 transposed, cached per weight tensor (inference only).  An entry is valid only for
    the tensor object it was computed from (ids and storage addresses get recycled) at the same version / device / address."""
    ent = _wsq_cache.get(id(weight))
    if ent is None or ent[0]() is not weight or ent[1] != _stamp(weight):
        _evict_dead(_wsq_cache)
        ent = (weakref.ref(weight), _stamp(weight), weight.detach().square().sum(dim=[2, 3]).t().contiguous())
        _wsq_cache[id(weight)] = ent
    return ent[2]


# ---- style prefetch (GPU inference) ----------------------------------------------------------------------------------
# The styles, demodulation coefficients and folded head weights of every layer depend only on `ws`, not on the
# activations: `prefetch_styles` computes them up front on a side stream (43 small launches per synthesis pass that would
# otherwise sit between the big convolution launches of the main stream) and the layers pick their result up by identity.
import os
import threading

_tls = threading.local()  # .prefetched: {id(affine module) -> (w data_ptr, result, event)} of the synthesis pass running on this thread
_side_streams = {}        # device -> stream (module level: generators are pickled / deep-copied, streams cannot be)


def _prefetch_table():
    return getattr(_tls, 'prefetched', None)


def side_stream(device):
    st = _side_streams.get(device)
    if st is None:
        from torch_utils import hip_plugin
        st = _side_streams[device] = hip_plugin.private_stream(device, 'style prefetch')      # not a pooled handle: it must never be the stream a pass is captured on
    return st


def _style_plan(block, ws_block):
    """(kind, module(s), w) in the order `SegSynthesisBlock.forward` consumes `ws_block` [N, num_conv + num_torgb, w_dim]."""
    ws_l, i = ws_block.unbind(dim=1), 0
    if block.in_channels != 0:
        yield 'conv', block.conv0, ws_l[i]; i += 1
    if not block.use_single_layer:
        yield 'conv', block.conv1, ws_l[i]; i += 1
    if block.is_last or block.architecture == 'skip':
        yield 'heads', block, ws_l[i]


def prefetch_styles(blocks_and_ws, side_stream):
    """Compute the styles, demodulation coefficients and folded head weights of all layers of `blocks_and_ws` =
    [(block, ws_block), ...] up front on `side_stream`: three batched launches (csrc/style.hip: all affines, all demodulations, all
    head foldings) where the batch form applies, one launch pair / one launch per layer otherwise.  The results live in a table owned
    by this call (thread-local, replaced per pass): two generators rendering on different threads never see each other's entries.
    Always pair with `finish_prefetch` in a try / finally.

    Why batched: as 43 separate nodes of a captured graph these 6 - 17 us launches run back to back BEFORE the first convolution of the
    pass — the graph executor does not start the convolution branch beside them (scripts/step_timeline.py: the first convolution
    started ~0.4 ms after the mapping network had finished) — so the pass pays for their count, not for their work."""
    _prefetched = _tls.prefetched = {}
    main = torch.cuda.current_stream()
    side_stream.wait_stream(main)
    with torch.cuda.stream(side_stream):
        plan = [(kind, mod, w) for block, ws_block in blocks_and_ws for kind, mod, w in _style_plan(block, ws_block)]
        conv_jobs, head_jobs = [], []
        batched = bool(plan) and _style_init() and not os.environ.get('IDE3D_NO_STYLE_BATCH')
        for kind, mod, w in plan:
            if kind == 'conv':
                a = mod.affine
                ok = (batched and _inference_on_gpu(w, a.weight, mod.weight) and w.ndim == 2 and w.stride(1) == 1 and w.shape[0] <= 8
                      and a.activation == 'linear' and a.bias is not None)
                if ok:
                    conv_jobs.append((mod, w))
            else:
                tr, ts = mod.torgb, mod.toseg
                ok = (batched and w.ndim == 2 and w.stride(1) == 1 and tr.weight.shape[2] == 1 and _inference_on_gpu(w, tr.weight, ts.weight))
                if ok:
                    head_jobs.append((mod, w))
        done = set()
        if conv_jobs:
            res = _style_plugin.style_demod_batch([(w, m.affine.weight, m.affine.bias, m.affine.weight_gain, m.affine.bias_gain, _wsq_t(m.weight))
                                                   for m, w in conv_jobs])
            if res is not None:
                ev = torch.cuda.Event(); ev.record(side_stream)
                for (m, w), r in zip(conv_jobs, res):
                    _prefetched[id(m.affine)] = (w.data_ptr(), r, ev); done.add(id(m.affine))
        if head_jobs:
            res = _style_plugin.fold_heads_batch([(w, b.torgb.affine.weight_gain,
                                                   b.torgb.affine.weight, b.torgb.affine.bias, b.torgb.weight.reshape(b.torgb.weight.shape[0], -1), b.torgb.weight_gain,
                                                   b.toseg.affine.weight, b.toseg.affine.bias, b.toseg.weight.reshape(b.toseg.weight.shape[0], -1), b.toseg.weight_gain)
                                                  for b, w in head_jobs])
            ev = torch.cuda.Event(); ev.record(side_stream)
            for (b, w), r in zip(head_jobs, res):
                _prefetched[id(b.torgb)] = (w.data_ptr(), r, ev); done.add(id(b.torgb))
        for kind, mod, w in plan:                       # whatever the batch form did not take: per layer, as before
            key = id(mod.affine) if kind == 'conv' else id(mod.torgb)
            if key in done:
                continue
            res = _styles_and_dcoefs(mod.affine, w, mod.weight, True) if kind == 'conv' else _folded_head_weights(mod.torgb, mod.toseg, w)
            if res is None:
                continue
            ev = torch.cuda.Event()
            ev.record(side_stream)
            _prefetched[key] = (w.data_ptr(), res, ev)


def finish_prefetch(side_stream):
    """Join the side stream (required before a hipGraph capture ends) and drop what was not consumed."""
    torch.cuda.current_stream().wait_stream(side_stream)
    _tls.prefetched = None


def _take_prefetched(key_module, w):
    ent = _prefetch_table().pop(id(key_module), None)
    if ent is None or ent[0] != w.data_ptr():
        return None
    main = torch.cuda.current_stream()
    main.wait_event(ent[2])
    for t in (ent[1] if isinstance(ent[1], tuple) else (ent[1],)):
        if t is not None:
            t.record_stream(main)            # allocated on the side stream, consumed (and freed) on this one
    return ent[1]


def _styles_and_dcoefs(affine, w, weight, demodulate):
    """(styles, dcoefs) of a modulated conv: ONE HIP launch (csrc/style.hip) in inference on device tensors,
    otherwise the PyTorch definition."""
    if _prefetch_table():
        hit = _take_prefetched(affine, w)
        if hit is not None:
            return hit
    if (_inference_on_gpu(w, affine.weight, weight) and w.ndim == 2 and w.stride(1) == 1 and affine.activation == 'linear'
            and affine.bias is not None and _style_init()):
        return _style_plugin.style_demod(w, affine.weight, affine.bias, affine.weight_gain, affine.bias_gain,
                                         _wsq_t(weight) if demodulate else None)
    styles = affine(w)
    return styles, (_demod_coefs(weight, styles) if demodulate else None)


# ---- f16x3 arithmetic: the producers' max |y| travels with the activation -------------------------------------------------------
# The fp16-split convolutions (hip_plugin.conv_arithmetic('f16x3')) scale their input patch by a power of two derived from a bound
# on |x| per image.  Every fused producer on the render path (convolution epilogue, FIR epilogue) can record max |y| per image for
# free while it stores y (`y_amax`, one atomic per workgroup); it rides on the tensor object as `_ide3d_amax` = (slots, the tensor's
# `_version`, its data pointer) and the consuming convolution picks it up only while all three still match (`_amax_of`).  A tensor without
# it (an external caller's input), one that was converted (another object) or modified in place since (another version) simply makes that
# launch run in bf16x6.
def _amax_wanted(x):
    if not (x.is_cuda and x.dtype == torch.float32) or torch.is_grad_enabled():
        return None
    from torch_utils import hip_plugin
    if hip_plugin.conv_arithmetic() != 'f16x3':
        return None
    arena = getattr(_tls, 'amax_arena', None)
    if arena is not None and arena[1] < arena[0].shape[0] and arena[0].shape[1] == x.shape[0] and arena[0].device == x.device:
        arena[1] += 1
        return arena[0][arena[1] - 1]
    return torch.zeros([x.shape[0], hip_plugin.AMAX_FLOATS], dtype=torch.float32, device=x.device)


class amax_arena:
    """`with amax_arena(n, device):` — one zero-filled tensor for the amax side outputs of every layer of a synthesis pass (one fill
    launch instead of one per layer; a no-op unless the f16x3 arithmetic is selected on a CUDA device)."""

    def __init__(self, n, device, layers=40):
        self.n, self.device, self.layers = n, torch.device(device), layers

    def __enter__(self):
        self.prev = getattr(_tls, 'amax_arena', None)
        _tls.amax_arena = None
        if self.device.type == 'cuda' and not torch.is_grad_enabled():
            from torch_utils import hip_plugin
            if hip_plugin.conv_arithmetic() == 'f16x3':
                _tls.amax_arena = [torch.zeros([self.layers, self.n, hip_plugin.AMAX_FLOATS], dtype=torch.float32, device=self.device), 0]
        return self

    def __exit__(self, *exc):
        _tls.amax_arena = self.prev
        return False


def _amax_of(x):
    """The producer's max |x| slots if `x` is still the tensor the producer returned: same object, same storage address and same
    version counter.  An in-place edit since (`x.mul_()`, `x.copy_()`, a forward hook) bumps `_version`, so a stale under-bound - which
    would overflow the fp16 pieces silently - is never handed on; the consuming launch then runs in bf16x6."""
    rec = getattr(x, '_ide3d_amax', None)
    if rec is None:
        return None
    amax, version, ptr = rec
    if x._version != version or x.data_ptr() != ptr:
        return None
    return amax


def _with_amax(y, amax):
    if amax is not None:
        y._ide3d_amax = (amax, y._version, y.data_ptr())
    return y


def _modconv_init():
    global _modconv_plugin
    if _modconv_plugin is None:
        _modconv_plugin = custom_ops.get_plugin(module_name='modconv_plugin', sources=['modconv.hip'])
    return True


def _inference_on_gpu(*tensors):
    x = tensors[0]
    if x.device.type != 'cuda' or x.dtype != torch.float32:
        return False
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        return False
    return True


@misc.profiled_function
def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


def _demod_coefs(weight, styles):
    """d[n, o] = rsqrt(sum_{i,k} (w[o,i,k] * s[n,i])^2 + 1e-8) without materialising per-sample weights."""
    if torch.is_grad_enabled() and weight.requires_grad:
        wsq_t = weight.square().sum(dim=[2, 3]).t()       # [I, O]
    else:
        wsq_t = _wsq_t(weight)       # sum_k w^2 only depends on the weights: cached, recomputed after any in-place update
    return torch.addmm(_eps_like(styles), styles.square(), wsq_t).rsqrt()     # [N, O]


_wsq_cache = {}
_noise_cache = {}
_wscale_cache = {}
_cat_cache = {}


def _cat_cached(a, b):
    """torch.cat([a, b]) of two parameters, formed once per (tensor objects, versions) in inference."""
    if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
        return torch.cat([a, b])
    ent = _cat_cache.get(id(a))
    if ent is None or ent[0]() is not a or ent[1]() is not b or ent[2] != (_stamp(a), _stamp(b)):
        _evict_dead(_cat_cache)
        ent = (weakref.ref(a), weakref.ref(b), (_stamp(a), _stamp(b)), torch.cat([a.detach(), b.detach()]))
        _cat_cache[id(a)] = ent
    return ent[3]


def _scaled_weight(weight, gain):
    """weight * gain (the equalised-learning-rate factor), formed once per (tensor object, version) in inference so that
    the packed copy inside the HIP workspace stays valid across calls."""
    ent = _wscale_cache.get(id(weight))
    if ent is None or ent[0]() is not weight or ent[1] != (_stamp(weight), float(gain)):
        _evict_dead(_wscale_cache)
        ent = (weakref.ref(weight), (_stamp(weight), float(gain)), (weight.detach() * gain).contiguous())
        _wscale_cache[id(weight)] = ent
    return ent[2]


def _scaled_const_noise(noise_const, noise_strength):
    """noise_const * noise_strength.  In inference both are fixed, so the product is formed once per (tensor objects,
    versions) instead of by one element-wise launch per layer and frame; with autograd it is the plain product."""
    if torch.is_grad_enabled() and (noise_strength.requires_grad or noise_const.requires_grad):
        return noise_const * noise_strength
    ent = _noise_cache.get(id(noise_const))
    if (ent is None or ent[0]() is not noise_const or ent[1]() is not noise_strength
            or ent[2] != (_stamp(noise_const), _stamp(noise_strength))):
        _evict_dead(_noise_cache)
        ent = (weakref.ref(noise_const), weakref.ref(noise_strength), (_stamp(noise_const), _stamp(noise_strength)),
               (noise_const * noise_strength).detach())
        _noise_cache[id(noise_const)] = ent
    return ent[3]
_eps_cache = {}


def _eps_like(t):
    k = (t.device, t.dtype)
    if k not in _eps_cache:
        _eps_cache[k] = torch.full([1], 1e-8, device=t.device, dtype=t.dtype)
    return _eps_cache[k]


@misc.profiled_function
def modulated_conv2d(
    x,                          # [N, Cin, H, W]
    weight,                     # [Cout, Cin, kh, kw]
    styles,                     # [N, Cin]
    noise           = None,     # tensor broadcastable to the output, added after the convolution
    up              = 1,
    down            = 1,
    padding         = 0,
    resample_filter = None,     # from upfirdn2d.setup_filter()
    demodulate      = True,
    flip_weight     = True,     # True = correlation (torch conv2d)
    fused_modconv   = True,     # one grouped conv over per-sample weights (reference :116-130) vs scale-conv-scale
    mode            = '2d',
    **unused,
):
    """Modulated convolution (reference inversion/networks.py:55-130)."""
    assert mode == '2d', 'only 2-D modulated convolutions are part of the render path'
    batch_size = x.shape[0]
    cout, cin, kh, kw = weight.shape

    # ---- MI355X inference path: one implicit-GEMM launch for stride-1 k in {1, 3} ----
    if (use_hip_modconv and up == 1 and down == 1 and kh == kw and kh in (1, 3) and padding == kh // 2 and flip_weight
            and _inference_on_gpu(x, weight, styles) and _modconv_init()):
        dcoefs = _demod_coefs(weight, styles) if demodulate else None
        y = _modconv_plugin.modconv2d(x.contiguous(), weight.contiguous(), styles.contiguous(), dcoefs,
                                      None, 0.0, None, 1, 0.0, 1.0, -1.0)
        if noise is not None:
            y = y.add_(noise)
        return y

    if x.dtype == torch.float16 and demodulate:   # pre-normalise to stay inside fp16 range (reference :76-80)
        weight = weight * (1 / np.sqrt(cin * kh * kw) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)

    if not fused_modconv:
        dcoefs = _demod_coefs(weight, styles) if demodulate else None
        x = x * styles.to(x.dtype).reshape(batch_size, -1, 1, 1)
        x = conv2d_resample.conv2d_resample(x=x, w=weight.to(x.dtype), f=resample_filter, up=up, down=down,
                                            padding=padding, flip_weight=flip_weight)
        if demodulate and noise is not None:
            x = fma.fma(x, dcoefs.to(x.dtype).reshape(batch_size, -1, 1, 1), noise.to(x.dtype))
        elif demodulate:
            x = x * dcoefs.to(x.dtype).reshape(batch_size, -1, 1, 1)
        elif noise is not None:
            x = x.add_(noise.to(x.dtype))
        return x

    # fused form: per-sample weights, one grouped convolution
    w = weight.unsqueeze(0) * styles.reshape(batch_size, 1, -1, 1, 1)           # [N, O, I, k, k]
    if demodulate:
        dcoefs = (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()                 # [N, O]
        w = w * dcoefs.reshape(batch_size, -1, 1, 1, 1)
    with misc.suppress_tracer_warnings():
        batch_size = int(batch_size)
    x = x.reshape(1, -1, *x.shape[2:])
    w = w.reshape(-1, cin, kh, kw)
    x = conv2d_resample.conv2d_resample(x=x, w=w.to(x.dtype), f=resample_filter, up=up, down=down, padding=padding,
                                        groups=batch_size, flip_weight=flip_weight)
    x = x.reshape(batch_size, -1, *x.shape[2:])
    if noise is not None:
        x = x.add_(noise)
    return x


def _modconv_bias_act(x, weight, styles, demodulate, noise2d, noise_strength, bias, act, gain, clamp, dcoefs=None):
    """Stride-1 modulated conv + noise + bias + activation in one HIP launch (inference only).
    Returns None when the fused kernel does not apply."""
    cout, cin, kh, kw = weight.shape
    if not (use_hip_modconv and kh == kw and kh in (1, 3) and act in ('linear', 'lrelu')
            and _inference_on_gpu(x, weight, styles, bias) and _modconv_init()):
        return None
    spec = bias_act.activation_funcs[act]
    if demodulate and dcoefs is None:
        dcoefs = _demod_coefs(weight, styles)
    amax = _amax_wanted(x) if kh == 3 else None
    xc = x.contiguous()
    return _with_amax(_modconv_plugin.modconv2d(
        xc, weight.contiguous(), styles.contiguous(), dcoefs, noise2d, noise_strength, bias,
        spec.cuda_idx, spec.def_alpha, gain, -1.0 if clamp is None else clamp,
        x_amax=(_amax_of(x) if xc is x else None), y_amax=amax), amax)


def _folded_head_weights(torgb, toseg, w):
    """Per-image 1x1 weights [N, Co_rgb + Co_seg, Cin, 1, 1] of the two heads with their styles folded in (ONE launch of
    csrc/style.hip), or None when the HIP path does not apply."""
    if not (w.ndim == 2 and w.stride(1) == 1 and torgb.weight.shape[2] == 1 and _inference_on_gpu(w, torgb.weight, toseg.weight)
            and _style_init()):
        return None
    return _style_plugin.fold_heads(w, torgb.affine.weight_gain,
                                    torgb.affine.weight, torgb.affine.bias, torgb.weight.reshape(torgb.weight.shape[0], -1), torgb.weight_gain,
                                    toseg.affine.weight, toseg.affine.bias, toseg.weight.reshape(toseg.weight.shape[0], -1), toseg.weight_gain)


def _adjacent_views(a, b):
    """The tensor `t` with a = t[:, :ca] and b = t[:, ca:] if a and b are exactly that (views of one [N, ca + cb, H, W] tensor), else None."""
    if a is None or b is None or a._base is None or a._base is not b._base:
        return None
    t = a._base
    if (t.ndim != 4 or a.ndim != 4 or b.ndim != 4 or not t.is_contiguous() or t.shape[1] != a.shape[1] + b.shape[1]
            or a.shape[0] != t.shape[0] or a.shape[2:] != t.shape[2:] or b.shape[0] != t.shape[0] or b.shape[2:] != t.shape[2:]
            or a.stride() != t.stride() or b.stride() != t.stride()
            or a.data_ptr() != t.data_ptr() or b.data_ptr() != t.data_ptr() + a.shape[1] * t.stride(1) * t.element_size()):
        return None
    return t


def _dual_head(x, torgb, toseg, w):
    """toRGB + toSeg of a dual-path block (reference networks.py:1109,1130) as ONE 1x1 implicit-GEMM launch.
    The two heads modulate with different styles, so the styles are folded into per-image weights
    [N, Cout_rgb + Cout_seg, Cin, 1, 1] (tiny) and the activation tensor x is read from HBM once.
    Inference on device tensors only; returns None otherwise."""
    if not (use_hip_modconv and torgb.weight.shape[2] == 1 and torgb.conv_clamp == toseg.conv_clamp
            and _inference_on_gpu(x, w, torgb.weight, toseg.weight) and _modconv_init()):
        return None
    n = x.shape[0]
    if w.shape[0] != n:
        return None
    wcat = _take_prefetched(torgb, w) if _prefetch_table() else None
    if wcat is None:
        wcat = _folded_head_weights(torgb, toseg, w)
    if wcat is None:
        s_rgb = torgb.affine(w) * torgb.weight_gain
        s_seg = toseg.affine(w) * toseg.weight_gain
        wr = torgb.weight[None, :, :, 0, 0] * s_rgb[:, None, :]           # [N, Co_rgb, Cin]
        ws = toseg.weight[None, :, :, 0, 0] * s_seg[:, None, :]
        wcat = torch.cat([wr, ws], dim=1)[:, :, :, None, None].contiguous()
    bias = _cat_cached(torgb.bias, toseg.bias)
    clamp = -1.0 if torgb.conv_clamp is None else torgb.conv_clamp
    y = _modconv_plugin.modconv2d(x.contiguous(), wcat, None, None, None, 0.0, bias, 1, 0.0, 1.0, clamp)
    co = torgb.weight.shape[0]
    return y[:, :co], y[:, co:]


_lowres_plugin = None


def _lowres_init():
    global _lowres_plugin
    if _lowres_plugin is None:
        _lowres_plugin = custom_ops.get_plugin(module_name='lowres_plugin', sources=['lowres.hip'])
    return True


def _hooked(*modules):
    """True when a forward (pre-)hook sits on one of `modules` (or globally): the hook wants that module's own call."""
    from torch.nn.modules import module as _m
    if _m._global_forward_hooks or _m._global_forward_pre_hooks:
        return True
    return any(m._forward_hooks or m._forward_pre_hooks for m in modules)


def lowres_group_forward(blocks, ws_per_block, noise_mode='const', force_fp32=False, **other_kwargs):
    """The leading dual-path blocks of a backbone — 4^2, 8^2, ... for as long as all images' maps of a layer fit one CU's LDS beside a weight
    slice — in ONE launch of csrc/lowres.hip (`ide3d_lowres_group`) instead of 2-4 launches per layer + 3 per pair of heads.
    Semantics: `SegSynthesisBlock.forward` of those blocks (reference inversion/networks.py:966-1139) with `SynthesisLayer` (:330-514)
    and `ToRGBLayer` (:670-713) in inference: const noise or none, fp32 blocks, 'skip' architecture, 3x3 layers of one width C, lrelu, the
    [1, 3, 3, 1] resample filter, products in the bf16x6 / bf16x3 arithmetic (`hip_plugin.conv_arithmetic`).  Anything else — autograd, a
    forward hook on one of the blocks or layers (viz/renderer.py:437), random noise, fp16 blocks, other arithmetics, `IDE3D_NO_LOWRES_GROUP` —
    keeps the per-layer path: returns None.
    Returns (x, img, seg, next_block, resume): the running tensors in front of `blocks[next_block]`; `resume` = x is already the output of that
    block's conv0 (the group may end in the middle of a block: `forward(..., _resume_after_conv0=True)`)."""
    if os.environ.get('IDE3D_NO_LOWRES_GROUP') or not use_hip_modconv or other_kwargs or noise_mode not in ('const', 'none'):
        return None
    if not blocks or len(blocks) != len(ws_per_block):
        return None
    b0 = blocks[0]
    ws0 = ws_per_block[0]
    if not (torch.is_tensor(ws0) and ws0.is_cuda and ws0.dtype == torch.float32) or (torch.is_grad_enabled() and ws0.requires_grad):
        return None
    if b0.in_channels != 0 or b0.use_single_layer or not hasattr(b0, 'const') or not _inference_on_gpu(b0.const, b0.conv1.weight):
        return None
    from torch_utils import hip_plugin
    if hip_plugin.conv_arithmetic() not in ('bf16x6', 'bf16x3'):
        return None
    C, n = b0.out_channels, ws0.shape[0]
    if C % 32 != 0 or n > 8:
        return None
    # the layers, in order, of every leading block that qualifies
    cand = []          # (block index, 'conv0' | 'conv1', layer)
    for bi, blk in enumerate(blocks):
        ok = (blk.architecture == 'skip' and not blk.use_single_layer and blk.out_channels == C and (bi == 0 or blk.in_channels == C)
              and not (blk.use_fp16 and not force_fp32) and not getattr(blk, 'skip_channels_last', False)
              and blk.torgb.conv_clamp == blk.toseg.conv_clamp and blk.torgb.weight.shape[2] == 1 and blk._filter_is_1331()
              and blk.resolution == b0.resolution << bi
              and not _hooked(blk, blk.conv1, blk.torgb, blk.toseg, *([blk.conv0] if bi else [])))
        if not ok:
            break
        for name in (('conv1',) if bi == 0 else ('conv0', 'conv1')):
            lay = getattr(blk, name)
            if not (lay.activation == 'lrelu' and lay.weight.shape == (C, C, 3, 3) and lay.padding == 1 and lay.up == (2 if name == 'conv0' else 1)
                    and _inference_on_gpu(ws0, lay.weight, lay.bias) and lay.weight.is_contiguous()
                    and (not lay.use_noise or noise_mode == 'none' or lay.noise_const.shape[-1] == blk.resolution)):
                ok = False
                break
            cand.append((bi, name, lay))
        if not ok:
            cand = [c for c in cand if c[0] < bi]
            break
    if not cand or not _lowres_init():
        return None
    ups = [lay.up for _, _, lay in cand]
    fit = _lowres_plugin.layers_supported(n, C, b0.resolution, ups)
    if fit < 1:
        return None
    cand = cand[:fit]
    last_bi, last_name, _ = cand[-1]
    resume = last_name == 'conv0'
    nheads = last_bi if resume else last_bi + 1                    # heads of every COMPLETE block
    if nheads > hip_plugin.LOWRES_MAX_HEADS or (nheads == 0):
        return None
    layers, heads = [], []
    for bi, name, lay in cand:
        blk, wsb = blocks[bi], ws_per_block[bi]
        w = wsb[:, (0 if (bi == 0 or name == 'conv0') else 1)]
        styles, dcoefs = _styles_and_dcoefs(lay.affine, w, lay.weight, True)
        noise = None
        if lay.use_noise and noise_mode == 'const':
            noise = _scaled_const_noise(lay.noise_const, lay.noise_strength)
        layers.append(dict(weight=lay.weight, styles=styles, dcoefs=dcoefs, noise=noise, bias=lay.bias, act_gain=lay.act_gain,
                           clamp=(-1.0 if lay.conv_clamp is None else lay.conv_clamp), up=lay.up,
                           head=(bi if (name == 'conv1' and bi < nheads) else -1)))
    for bi in range(nheads):
        blk, wsb = blocks[bi], ws_per_block[bi]
        w = wsb[:, blk.num_conv]
        wcat = _take_prefetched(blk.torgb, w) if _prefetch_table() else None
        if wcat is None:
            wcat = _folded_head_weights(blk.torgb, blk.toseg, w)
        if wcat is None:
            return None
        heads.append(dict(w=wcat, bias=_cat_cached(blk.torgb.bias, blk.toseg.bias), clamp=(-1.0 if blk.torgb.conv_clamp is None else blk.torgb.conv_clamp)))
    x, skips = _lowres_plugin.group(b0.const.detach(), layers, heads, b0.resample_filter)
    co = blocks[0].torgb.weight.shape[0]
    skip = skips[-1]
    return x, skip[:, :co], skip[:, co:], nheads, resume


@persistence.persistent_class
class FullyConnectedLayer(torch.nn.Module):
    """Equalised-learning-rate dense layer (reference networks.py:136-165)."""

    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def effective(self, dtype=torch.float32):
        """(weight, bias) with the runtime gains folded in."""
        w = self.weight.to(dtype) * self.weight_gain
        b = self.bias
        if b is not None:
            b = b.to(dtype)
            if self.bias_gain != 1:
                b = b * self.bias_gain
        return w, b

    def forward(self, x):
        if _inference_on_gpu(x, self.weight, self.bias):
            # gains folded into the parameters once per (tensor, version): two element-wise launches less per layer
            w = _scaled_weight(self.weight, self.weight_gain)
            b = None if self.bias is None else (_scaled_weight(self.bias, self.bias_gain) if self.bias_gain != 1 else self.bias)
        else:
            w, b = self.effective(x.dtype)
        if self.activation == 'linear' and b is not None:
            return torch.addmm(b.unsqueeze(0), x, w.t())
        x = x.matmul(w.t())
        return bias_act.bias_act(x, b, act=self.activation)


@persistence.persistent_class
class Conv2dLayer(torch.nn.Module):
    """Plain (un-modulated) convolution with optional resampling (reference networks.py:170-226)."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False, trainable=True, **unused):
        super().__init__()
        self.activation = activation
        self.up = up
        self.down = down
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        memory_format = torch.channels_last if channels_last else torch.contiguous_format
        weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format)
        bias = torch.zeros([out_channels]) if bias else None
        if trainable:
            self.weight = torch.nn.Parameter(weight)
            self.bias = torch.nn.Parameter(bias) if bias is not None else None
        else:
            self.register_buffer('weight', weight)
            if bias is not None:
                self.register_buffer('bias', bias)
            else:
                self.bias = None

    def forward(self, x, gain=1):
        k = self.weight.shape[-1]
        if (self.up == 1 and self.down == 1 and k in (1, 3) and self.activation in ('linear', 'lrelu') and use_hip_modconv
                and _inference_on_gpu(x, self.weight, self.bias) and _modconv_init()):
            # MI355X inference: conv + bias + activation in ONE launch of the MFMA kernel (csrc/modconv.hip, no modulation)
            spec = bias_act.activation_funcs[self.activation]
            act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
            return _modconv_plugin.modconv2d(x.contiguous(), _scaled_weight(self.weight, self.weight_gain), None, None, None, 0.0,
                                             self.bias, spec.cuda_idx, spec.def_alpha, self.act_gain * gain,
                                             -1.0 if act_clamp is None else act_clamp)
        if (self.up == 1 and self.down == 2 and k in (1, 3) and self.activation in ('linear', 'lrelu') and use_hip_modconv
                and self.resample_filter.ndim == 2 and _inference_on_gpu(x, self.weight, self.bias) and _modconv_init()):
            # down-sampling layer, same decomposition as conv2d_resample.py:73-78,95-103: k = 1 -> FIR + decimation, then the
            # 1x1 conv at low resolution; k = 3 -> FIR at full resolution, then the stride-2 conv (mode 1 of the MFMA kernel)
            spec = bias_act.activation_funcs[self.activation]
            act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
            fw = self.resample_filter.shape[-1]
            p0, p1 = self.padding + (fw - self.down + 1) // 2, self.padding + (fw - self.down) // 2
            if k == 1:
                y = upfirdn2d.upfirdn2d(x, self.resample_filter, down=2, padding=[p0, p1, p0, p1])
                mode = 0
            else:
                y = upfirdn2d.upfirdn2d(x, self.resample_filter, padding=[p0, p1, p0, p1])
                mode = 1
            return _modconv_plugin.modconv2d(y.contiguous(), _scaled_weight(self.weight, self.weight_gain), None, None, None, 0.0,
                                             self.bias, spec.cuda_idx, spec.def_alpha, self.act_gain * gain,
                                             -1.0 if act_clamp is None else act_clamp, mode=mode)
        w = self.weight * self.weight_gain
        b = self.bias.to(x.dtype) if self.bias is not None else None
        x = conv2d_resample.conv2d_resample(x=x, w=w.to(x.dtype), f=self.resample_filter, up=self.up, down=self.down,
                                            padding=self.padding, flip_weight=(self.up == 1))
        act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return bias_act.bias_act(x, b, act=self.activation, gain=self.act_gain * gain, clamp=act_clamp)


@persistence.persistent_class
class MappingNetwork(torch.nn.Module):
    """z (+ camera label c) -> ws (reference networks.py:246-325)."""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.995, **unused):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.num_ws, self.num_layers, self.w_avg_beta = num_ws, num_layers, w_avg_beta
        if embed_features is None:
            embed_features = w_dim
        if c_dim == 0:
            embed_features = 0
        if layer_features is None:
            layer_features = w_dim
        widths = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', FullyConnectedLayer(widths[idx], widths[idx + 1], activation=activation,
                                                          lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def _forward_hip(self, z, c, truncation_psi, truncation_cutoff):
        """The whole network in ONE launch of csrc/mapping.hip (inference on device tensors, batch <= 8), or None."""
        global _mapping_plugin
        if self.z_dim == 0 or self.num_ws is None or not (z.device.type == 'cuda' and not torch.is_grad_enabled()):
            return None
        # What the one-launch kernel requires of the layers (one activation, one lr_multiplier, fp32) is a property of the parameter OBJECTS:
        # checked once per set of objects, not on every call (the drop-in loop's host time sits on its critical path: bench.py dropin_eager_b1)
        layers = [getattr(self, f'fc{i}') for i in range(self.num_layers)]
        embed = self.embed if self.c_dim > 0 else None
        ident = tuple(id(l.weight) for l in layers) + (id(embed.weight) if embed is not None else 0, layers[0].weight.dtype, layers[0].weight.device, layers[-1].weight.device,
                                                       z.device, z.shape[0])
        plan = getattr(self, '_hip_plan', None)
        trusted = plan is not None and plan[0] == ident
        if not trusted:
            if any(l.activation != 'lrelu' or l.weight.dtype != torch.float32 or l.bias_gain != layers[0].bias_gain
                   or not math.isclose(l.weight_gain * math.sqrt(l.weight.shape[1]), l.bias_gain, rel_tol=1e-6) for l in layers):      # one lr_multiplier
                return None
            if embed is not None and embed.activation != 'linear':
                return None
            if _mapping_plugin is None:
                _mapping_plugin = custom_ops.get_plugin(module_name='mapping_plugin', sources=['mapping.hip'])
            if not _mapping_plugin.supports(z.shape[0], self.z_dim, 0 if embed is None else embed.weight.shape[0], [l.weight.shape[0] for l in layers]):
                return None
        if embed is not None and c is None:
            return None
        if truncation_psi != 1 and self.w_avg_beta is None:
            return None
        spec = bias_act.activation_funcs['lrelu']
        zc = z.to(torch.float32).contiguous()
        cc = None if embed is None else c.to(torch.float32).contiguous()
        ws = _mapping_plugin.mapping(
            zc, cc, None if embed is None else embed.weight, None if embed is None else embed.bias,
            1.0 if embed is None else embed.weight_gain, 1.0 if embed is None else embed.bias_gain,
            [l.weight for l in layers], [l.bias for l in layers], layers[0].bias_gain, spec.def_alpha, spec.def_gain,
            self.num_ws, self.w_avg if (self.w_avg_beta is not None) else None, float(truncation_psi), truncation_cutoff, trusted=trusted)
        if not trusted:
            object.__setattr__(self, '_hip_plan', (ident,))          # plain attribute: never a buffer / parameter, never pickled state that matters
        return ws

    def forward(self, z=None, c=None, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False, styles=None,
                **unused_kwargs):
        if styles is not None:
            return styles
        if z is not None and not self.training:
            misc.assert_shape(z, [None, self.z_dim])
            if self.c_dim > 0:
                misc.assert_shape(c, [None, self.c_dim])
            ws = self._forward_hip(z, c, truncation_psi, truncation_cutoff)
            if ws is not None:
                return ws
        x = None
        with torch.autograd.profiler.record_function('input'):
            if self.z_dim > 0:
                misc.assert_shape(z, [None, self.z_dim])
                x = normalize_2nd_moment(z.to(torch.float32))
            if self.c_dim > 0:
                misc.assert_shape(c, [None, self.c_dim])
                y = normalize_2nd_moment(self.embed(c.to(torch.float32)))
                x = torch.cat([x, y], dim=1) if x is not None else y
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x)
        if self.training and self.w_avg_beta is not None and not skip_w_avg_update:
            self._track_w_avg(x)
        if self.num_ws is not None:
            x = x[:, None, :].repeat(1, self.num_ws, 1)                      # one copy of w per synthesis layer
        return self._truncate(x, truncation_psi, truncation_cutoff)

    def _track_w_avg(self, w):
        """Training only (reference networks.py:308-311): w_avg <- beta * w_avg + (1 - beta) * batch mean of w."""
        batch_mean = w.detach().mean(dim=0)
        self.w_avg.copy_(torch.lerp(batch_mean, self.w_avg, self.w_avg_beta))

    def _truncate(self, ws, psi, cutoff):
        """Truncation trick (reference networks.py:318-324): pull w towards w_avg by 1 - psi, on the first `cutoff` layers only when given."""
        if psi == 1:
            return ws
        assert self.w_avg_beta is not None, 'truncation needs the tracked w_avg'
        if cutoff is None or self.num_ws is None:
            return torch.lerp(self.w_avg, ws, psi)
        ws[:, :cutoff] = torch.lerp(self.w_avg, ws[:, :cutoff], psi)
        return ws


@persistence.persistent_class
class SynthesisLayer(torch.nn.Module):
    """Modulated 3x3 conv (+ optional x2 up-sampling) + noise + bias + lrelu (reference networks.py:330-514,
    `upsample_mode='default'`, 2-D)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True,
                 activation='lrelu', resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False, **unused_kwargs):
        super().__init__()
        self.resolution = resolution
        self.up = up
        self.use_noise = use_noise
        self.activation = activation
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        memory_format = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format))
        if use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1, input_noise=None, **unused_kwargs):
        assert noise_mode in ['random', 'const', 'none']
        styles, dcoefs_pre = _styles_and_dcoefs(self.affine, w, self.weight, True)
        if styles.size(0) < x.size(0):
            dcoefs_pre = None
            assert x.size(0) % styles.size(0) == 0
            styles = styles.repeat_interleave(x.size(0) // styles.size(0), dim=0)

        noise = None
        const_noise = False
        if self.use_noise:
            if input_noise is not None:
                noise = input_noise * self.noise_strength
            elif noise_mode == 'random':
                noise = torch.randn([x.shape[0], 1, self.up * x.shape[2], self.up * x.shape[3]], device=x.device) * self.noise_strength
            elif noise_mode == 'const':
                const_noise = self.noise_const.shape[-1] >= self.up * x.shape[3]
                noise = _scaled_const_noise(self.noise_const, self.noise_strength)
                if not const_noise:
                    noise = noise.repeat(1, self.up * x.shape[3] // noise.shape[-1])

        act_gain = self.act_gain * gain
        act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None

        if (self.up == 2 and use_hip_modconv and self.weight.shape[2] == 3 and self.padding == 1
                and _inference_on_gpu(x, self.weight, styles, self.bias) and _modconv_init()):
            # up-sampling layer, MI355X inference path (same strategy as conv2d_resample.py:112-129): transposed
            # 3x3 stride-2 conv as a parity-class implicit GEMM (demodulation fused), then the 4x4 FIR with gain 4.
            dcoefs = dcoefs_pre if dcoefs_pre is not None else _demod_coefs(self.weight, styles)
            xc = x.contiguous()
            y = _modconv_plugin.modconv2d(xc, self.weight.contiguous(), styles.contiguous(), dcoefs,
                                          None, 0.0, None, 1, 0.0, 1.0, -1.0, mode=2, x_amax=(_amax_of(x) if xc is x else None), pad_rows=True)
            spec = bias_act.activation_funcs[self.activation]
            if self.activation in ('linear', 'lrelu') and (noise is None or (noise.ndim == 2 and noise.shape == (2 * x.shape[2], 2 * x.shape[3]))):
                # FIR + noise + bias + lrelu in one launch (ide3d_upfirdn2d_ex)
                amax = _amax_wanted(y)
                return _with_amax(_upfirdn_plugin().upfirdn2d_ex(y, self.resample_filter, 1, 1, 1, 1, 1, 1, 1, 1, False, 4.0,
                                                                 noise=noise, noise_strength=1.0, bias=self.bias, act=spec.cuda_idx,
                                                                 alpha=spec.def_alpha, act_gain=act_gain,
                                                                 clamp=(-1.0 if act_clamp is None else act_clamp), y_amax=amax), amax)
            y = upfirdn2d.upfirdn2d(y, self.resample_filter, padding=[1, 1, 1, 1], gain=4)
            if noise is not None:
                y = y.add_(noise)
            return bias_act.bias_act(y, self.bias.to(x.dtype), act=self.activation, gain=act_gain, clamp=act_clamp)
        if self.up == 1:
            # single-launch path: conv + const/absent noise + bias + activation
            if noise is None or (const_noise and input_noise is None and noise.shape == x.shape[2:]):
                # `noise` already carries noise_strength (device-side product: no host sync)
                y = _modconv_bias_act(x, self.weight, styles, True, noise, 1.0,
                                      self.bias.to(x.dtype), self.activation, act_gain, act_clamp, dcoefs=dcoefs_pre)
                if y is not None:
                    return y
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, noise=noise, up=self.up, padding=self.padding,
                             resample_filter=(self.resample_filter if self.up > 1 else None), flip_weight=(self.up == 1),
                             fused_modconv=(fused_modconv and not _inference_on_gpu(x, self.weight, styles)))
        return bias_act.bias_act(x, self.bias.to(x.dtype), act=self.activation, gain=act_gain, clamp=act_clamp)


@persistence.persistent_class
class ToRGBLayer(torch.nn.Module):
    """1x1 modulated conv without demodulation -> image / segmentation head (reference networks.py:670-713, w_dim > 0)."""

    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False, **unused):
        super().__init__()
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        memory_format = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))

    def forward(self, x, w, fused_modconv=True):
        styles = self.affine(w) * self.weight_gain
        if x.size(0) > styles.size(0):
            assert x.size(0) % styles.size(0) == 0
            styles = styles.repeat_interleave(x.size(0) // styles.size(0), dim=0)
        y = _modconv_bias_act(x, self.weight, styles, False, None, 0.0, self.bias.to(x.dtype), 'linear', 1.0, self.conv_clamp)
        if y is not None:
            return y
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, demodulate=False, fused_modconv=fused_modconv)
        return bias_act.bias_act(x, self.bias.to(x.dtype), clamp=self.conv_clamp)


@persistence.persistent_class
class SegSynthesisBlock(torch.nn.Module):
    """Dual-path synthesis block: one conv trunk, an image head (torgb) and a semantic head (toseg) that share
    the same w (reference networks.py:966-1139; 'skip' and 'orig' architectures, default up-sampling)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, seg_channels, is_last,
                 architecture='skip', resample_filter=[1, 3, 3, 1], conv_clamp=None, use_fp16=False,
                 fp16_channels_last=False, use_single_layer=False, disable_upsample=False, **layer_kwargs):
        assert architecture in ['orig', 'skip']
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.w_dim, self.resolution = w_dim, resolution
        self.img_channels, self.seg_channels = img_channels, seg_channels
        self.is_last, self.architecture = is_last, architecture
        self.use_fp16 = use_fp16
        self.channels_last = (use_fp16 and fp16_channels_last)
        self.use_single_layer = use_single_layer
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.num_conv = self.num_torgb = self.num_toseg = 0
        layer_name = layer_kwargs.pop('layer_name', 'training.networks.SynthesisLayer')

        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        else:
            self.conv0 = util.construct_class_by_name(
                class_name=layer_name, in_channels=in_channels, out_channels=out_channels, w_dim=w_dim,
                resolution=resolution, up=(1 if disable_upsample else 2), resample_filter=resample_filter,
                conv_clamp=conv_clamp, channels_last=self.channels_last, **layer_kwargs)
            self.num_conv += 1
        if not use_single_layer:
            self.conv1 = util.construct_class_by_name(
                class_name=layer_name, in_channels=out_channels, out_channels=out_channels, w_dim=w_dim,
                resolution=resolution, conv_clamp=conv_clamp, channels_last=self.channels_last, **layer_kwargs)
            self.num_conv += 1
        if is_last or architecture == 'skip':
            self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp, channels_last=self.channels_last)
            self.num_torgb += 1
            self.toseg = ToRGBLayer(out_channels, seg_channels, w_dim=w_dim, conv_clamp=conv_clamp, channels_last=self.channels_last)
            self.num_toseg += 1

    def _merge_skip(self, skip, x):
        """Bring the running skip image to the resolution of x."""
        if skip is None:
            return None
        if skip.size(-1) * 2 == x.size(-1):
            return upfirdn2d.upsample2d(skip, self.resample_filter)
        if skip.size(-1) == x.size(-1):
            return skip
        raise NotImplementedError

    def _accumulate(self, lo, cur, y):
        """skip = upsample2d(lo) + y  (one HIP launch when `lo` is the deferred low-resolution skip image).  A block flagged
        `skip_channels_last` (the last tri-plane block: its skip images ARE the tri-planes the ray-marcher gathers from) gets
        the result in channels_last memory format straight from the kernel (csrc/resample.hip) instead of NCHW + a transpose."""
        if lo is not None:
            if (getattr(self, 'skip_channels_last', False) and lo.shape[1] % 4 == 0 and self.resample_filter.shape == (4, 4)
                    and _inference_on_gpu(lo, y) and self._filter_is_1331() and _resample_init()):
                return _resample_plugin.skip_upsample_add_cl(lo, y)
            f = self.resample_filter
            return _upfirdn_plugin().upfirdn2d_ex(lo, f, 2, 2, 1, 1, 2, 1, 2, 1, False, 4.0, add=y)
        return cur.add_(y) if cur is not None else y

    def _filter_is_1331(self):
        """The fused channels-last skip kernel hard-codes the [1,3,3,1] x [1,3,3,1] / 64 filter (checked once per module)."""
        ok = getattr(self, '_f1331', None)
        if ok is None:
            ref = upfirdn2d.setup_filter([1, 3, 3, 1])
            ok = self._f1331 = bool(torch.equal(self.resample_filter.detach().cpu(), ref))
        return ok

    def forward(self, x, img, seg, ws, force_fp32=False, fused_modconv=None, block_noise=None, disable_rgb=False, **layer_kwargs):
        misc.assert_shape(ws, [None, self.num_conv + self.num_torgb, self.w_dim])
        resume_after_conv0 = bool(layer_kwargs.pop('_resume_after_conv0', False))
        w_iter = iter(ws.unbind(dim=1))
        dtype = torch.float16 if self.use_fp16 and not force_fp32 else torch.float32
        memory_format = torch.channels_last if self.channels_last and not force_fp32 else torch.contiguous_format
        # fp16 blocks (reference :1058-1060: the high-resolution blocks of a released pickle store activations in fp16 with
        # conv_clamp = 256) in MI355X inference: the HIP kernels compute in fp32 — exact products on the fp32 MFMA path, the same
        # clamp fused in the epilogue — and the block's output activation is rounded to fp16 like the reference's storage format, so
        # the tensors that cross block boundaries (and that viewer hooks see) have the reference's dtype at >= its precision.
        out_dtype = dtype
        if dtype == torch.float16 and ws.is_cuda and not torch.is_grad_enabled() and use_hip_modconv:
            dtype, memory_format = torch.float32, torch.contiguous_format
        if fused_modconv is None:
            with misc.suppress_tracer_warnings():
                fused_modconv = (not self.training) and (dtype == torch.float32 or int(ws.shape[0]) == 1)

        if self.in_channels == 0:
            x = self.const.to(dtype=dtype, memory_format=memory_format)
            x = x.unsqueeze(0).expand(ws.shape[0], *x.size())
            if not self.use_single_layer:
                layer_kwargs['input_noise'] = block_noise[:, 1:2] if block_noise is not None else None
                x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
        else:
            x = x.to(dtype=dtype, memory_format=memory_format)
            layer_kwargs['input_noise'] = block_noise[:, 0:1] if block_noise is not None else None
            if resume_after_conv0:
                next(w_iter)          # x already is conv0's output (lowres_group_forward ended inside this block)
            else:
                x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            if not self.use_single_layer:
                layer_kwargs['input_noise'] = block_noise[:, 1:2] if block_noise is not None else None
                x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)

        w_shared = next(w_iter) if (self.is_last or self.architecture == 'skip') else None
        img_lo = seg_lo = None
        heads_follow = (self.is_last or self.architecture == 'skip') and not disable_rgb
        if heads_follow and _inference_on_gpu(x) and img is not None and seg is not None and img.size(-1) * 2 == x.size(-1):
            img_lo, seg_lo, img, seg = img, seg, None, None       # defer: upsample + add in one launch (_accumulate)
        else:
            img = self._merge_skip(img, x)
            seg = self._merge_skip(seg, x)
        if self.is_last or self.architecture == 'skip':
            if disable_rgb:
                img = seg = None
            else:
                heads = _dual_head(x, self.torgb, self.toseg, w_shared)
                if heads is not None:
                    y, y_seg = heads              # one launch: x is read once for both heads
                else:
                    y = self.torgb(x, w_shared, fused_modconv=fused_modconv).to(dtype=torch.float32, memory_format=torch.contiguous_format)
                    y_seg = self.toseg(x, w_shared, fused_modconv=fused_modconv).to(dtype=torch.float32, memory_format=torch.contiguous_format)
                lo_both, y_both = _adjacent_views(img_lo, seg_lo), _adjacent_views(y, y_seg)
                if (lo_both is not None and y_both is not None and img is None and seg is None and not getattr(self, 'skip_channels_last', False)
                        and not self.is_last and not os.environ.get('IDE3D_NO_SKIP_MERGE')):          # (the network's outputs stay dense tensors of their own, like the reference's)
                    # both skip images in ONE up-sample + add launch: they are channel ranges of one tensor (the dual head's output,
                    # the previous block's accumulation) all the way through the backbone
                    both = self._accumulate(lo_both, None, y_both)
                    img, seg = both[:, :y.shape[1]], both[:, y.shape[1]:]
                else:
                    img = self._accumulate(img_lo, img, y)
                    seg = self._accumulate(seg_lo, seg, y_seg)
        assert x.dtype == dtype
        if out_dtype != dtype:
            x = x.to(out_dtype)
        assert img is None or img.dtype == torch.float32
        assert seg is None or seg.dtype == torch.float32
        return x, img, seg
