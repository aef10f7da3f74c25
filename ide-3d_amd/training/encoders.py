"""Image / segmentation -> latent encoders of the interactive editing loop (reference inversion/networks.py:1507-1665,
used by apps/train_hybrid_encoder.py:208 and Painter/run_UI.py:193-199): the inverse half of "edit -> re-render".

`HybridEncoder(img, seg) -> ws [B, n_latents_geo + n_latents_app, w_dim]`: two identical residual conv towers (one on
the RGB image, one on the 19-channel segmentation), each halving the resolution down to 4x4 and projecting with a 4x4
convolution.  Every convolution is a `Conv2dLayer` (training/networks.py), so on the GPU the stride-1 convolutions run on
the MFMA kernel of csrc/modconv.hip, the low-pass filter of the down-sampling layers on csrc/upfirdn2d.hip and the
bias / activation on csrc/bias_act.hip; module and parameter names equal the reference's, state dicts interchange.
"""

import math

import torch

from torch_utils import persistence
from training.networks import Conv2dLayer


def tower_channels(resolution: int) -> int:
    """Feature width at a given resolution (reference table, inversion/networks.py:1564-1574)."""
    return {4: 512, 8: 512, 16: 512, 32: 512, 64: 256, 128: 128, 256: 64, 512: 32, 1024: 16}[resolution]


@persistence.persistent_class
class EncoderResBlock(torch.nn.Module):
    """3x3 conv, 3x3 down-2 conv and a 1x1 down-2 skip, summed and scaled by 1/sqrt(2) (reference :1507-1521)."""

    def __init__(self, in_channel, out_channel, blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        self.conv1 = Conv2dLayer(in_channel, in_channel, 3, activation='lrelu')
        self.conv2 = Conv2dLayer(in_channel, out_channel, 3, down=2, activation='lrelu')
        self.skip = Conv2dLayer(in_channel, out_channel, 1, down=2, activation='linear', bias=False)

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return (y + self.skip(x)) * (1 / math.sqrt(2))


@persistence.persistent_class
class EqualConv2d(torch.nn.Module):
    """Plain conv with equalised learning rate: weight * 1/sqrt(fan_in) at run time (reference :1524-1556)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride, self.padding = stride, padding
        self.bias = torch.nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, x):
        return torch.nn.functional.conv2d(x, self.weight * self.scale, bias=self.bias, stride=self.stride, padding=self.padding)

    def extra_repr(self):
        o, i, k, _ = self.weight.shape
        return f'{i}, {o}, {k}, stride={self.stride}, padding={self.padding}'


def _tower(input_dim: int, size: int):
    """1x1 stem at `size`, then residual blocks down to 4x4 -> (Sequential, channels at 4x4)."""
    layers = [Conv2dLayer(input_dim, tower_channels(size), 1)]
    ch = tower_channels(size)
    res = size
    while res > 4:
        res //= 2
        layers.append(EncoderResBlock(ch, tower_channels(res)))
        ch = tower_channels(res)
    return torch.nn.Sequential(*layers), ch


@persistence.persistent_class
class Encoder(torch.nn.Module):
    """Single-tower encoder (reference :1559-1601)."""

    def __init__(self, size, n_latents, w_dim=512, add_dim=0, input_dim=3, **unused):
        super().__init__()
        self.w_dim, self.add_dim, self.n_latents = w_dim, add_dim, n_latents
        self.convs, ch = _tower(input_dim, size)
        self.projector = EqualConv2d(ch, n_latents * w_dim + add_dim, 4, padding=0, bias=False)

    def forward(self, x):
        out = self.projector(self.convs(x))
        if self.add_dim == 0:
            return out.view(len(x), self.n_latents, self.w_dim)
        # like the reference, the extra head is read from the last two channels whatever add_dim says
        return out[:, :-2].view(len(x), self.n_latents, self.w_dim), out[:, -2:].view(len(x), self.add_dim)


@persistence.persistent_class
class HybridEncoder(torch.nn.Module):
    """Appearance tower on the image + geometry tower on the segmentation (reference :1604-1665); output rows are
    [geometry latents | appearance latents]."""

    def __init__(self, size, n_latents_app, n_latents_geo, w_dim=512, add_dim=0, input_img_dim=3, input_seg_dim=19, **unused):
        super().__init__()
        self.w_dim, self.add_dim = w_dim, add_dim
        self.n_latents_app, self.n_latents_geo = n_latents_app, n_latents_geo
        self.convs_img, ch = _tower(input_img_dim, size)
        self.projector_img = EqualConv2d(ch, n_latents_app * w_dim + add_dim, 4, padding=0, bias=False)
        self.convs_seg, ch = _tower(input_seg_dim, size)
        self.projector_seg = EqualConv2d(ch, n_latents_geo * w_dim, 4, padding=0, bias=False)

    def forward(self, img, seg):
        b = img.size(0)
        out_img = self.projector_img(self.convs_img(img))
        out_seg = self.projector_seg(self.convs_seg(seg)).view(b, self.n_latents_geo, self.w_dim)
        if self.add_dim == 0:
            return torch.cat([out_seg, out_img.view(b, self.n_latents_app, self.w_dim)], 1)
        ws_img, extra = out_img[:, :-2], out_img[:, -2:]
        return torch.cat([out_seg, ws_img.view(b, self.n_latents_app, self.w_dim)], 1), extra.view(b, self.add_dim)
