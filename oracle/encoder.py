"""CPU ORACLE for the image / segmentation encoders — TEST INFRASTRUCTURE ONLY (rules: header of oracle/ops.py).

Functional restatement of `HybridEncoder.forward` / `Encoder.forward` (inversion/networks.py:1507-1665) on a state
dict, written against the op oracles (`oracle/ops.py` or `oracle/fast_ops.py`):
  Conv2dLayer (inversion/networks.py:169-226)  w * 1/sqrt(fan_in); down=2 -> upfirdn2d with the [1,3,3,1] filter and
      padding p + (fw - down + 1) // 2 / p + (fw - down) // 2 (conv2d_resample.py:73-78,100-103), then a stride-2
      convolution without padding; then bias_act(act, default gain).
  EncoderResBlock (:1507-1521)  (conv2(conv1(x)) + skip(x)) / sqrt(2)
  EqualConv2d (:1524-1556)      conv2d(x, w / sqrt(fan_in), no padding)
Pinned by tests/test_oracle_golden.py::test_encoder_oracle against a reference run (tests/golden/encoder.npz).
"""

import math

import torch
import torch.nn.functional as F

from . import ops as precise_ops

RESAMPLE = [1, 3, 3, 1]


def conv2d_layer(sd, prefix, x, k, down=1, act='linear', ops=precise_ops):
    w = sd[prefix + '.weight'].float()
    w = w * (1 / math.sqrt(w.shape[1] * k * k))
    pad = k // 2
    if down == 1:
        y = F.conv2d(x, w, padding=pad)
    else:
        fw = len(RESAMPLE)
        p0, p1 = pad + (fw - down + 1) // 2, pad + (fw - down) // 2
        y = ops.upfirdn2d(x, ops.setup_filter(RESAMPLE), padding=[p0, p1, p0, p1])
        y = F.conv2d(y, w, stride=down)
    b = sd.get(prefix + '.bias')
    return ops.bias_act(y, None if b is None else b.float(), act=act)


def res_block(sd, prefix, x, ops=precise_ops):
    y = conv2d_layer(sd, prefix + '.conv1', x, 3, act='lrelu', ops=ops)
    y = conv2d_layer(sd, prefix + '.conv2', y, 3, down=2, act='lrelu', ops=ops)
    s = conv2d_layer(sd, prefix + '.skip', x, 1, down=2, act='linear', ops=ops)
    return (y + s) / math.sqrt(2)


def tower(sd, prefix, x, ops=precise_ops):
    x = conv2d_layer(sd, prefix + '.0', x, 1, ops=ops)
    i = 1
    while f'{prefix}.{i}.conv1.weight' in sd:
        x = res_block(sd, f'{prefix}.{i}', x, ops=ops)
        i += 1
    return x


def equal_conv(sd, prefix, x):
    w = sd[prefix + '.weight'].float()
    return F.conv2d(x, w * (1 / math.sqrt(w.shape[1] * w.shape[2] ** 2)))


def hybrid_encoder(sd, img, seg, n_latents_app, n_latents_geo, w_dim, ops=precise_ops):
    """-> ws [B, n_latents_geo + n_latents_app, w_dim] (add_dim = 0)."""
    b = img.shape[0]
    out_img = equal_conv(sd, 'projector_img', tower(sd, 'convs_img', img.float(), ops)).reshape(b, n_latents_app, w_dim)
    out_seg = equal_conv(sd, 'projector_seg', tower(sd, 'convs_seg', seg.float(), ops)).reshape(b, n_latents_geo, w_dim)
    return torch.cat([out_seg, out_img], 1)


def encoder(sd, x, n_latents, w_dim, add_dim=0, ops=precise_ops):
    out = equal_conv(sd, 'projector', tower(sd, 'convs', x.float(), ops))
    if add_dim == 0:
        return out.reshape(len(x), n_latents, w_dim)
    return out[:, :-2].reshape(len(x), n_latents, w_dim), out[:, -2:].reshape(len(x), add_dim)
