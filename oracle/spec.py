"""Architecture record used by the oracle (same fields as `training.triplane.GeneratorSpec`; duplicated so that the
oracle never imports product code).  TEST INFRASTRUCTURE ONLY."""

import dataclasses
from typing import Dict, List, Optional

import numpy as np


@dataclasses.dataclass
class Spec:
    z_dim: int = 512
    c_dim: int = 25
    w_dim: int = 512
    img_resolution: int = 512
    img_channels: int = 3
    seg_channels: int = 19
    mapping_layers: int = 8
    channel_base: int = 32768
    channel_max: int = 512
    plane_resolution: int = 256
    plane_channels: int = 32
    render_size: int = 64
    feature_channels: int = 32
    decoder_hidden: int = 64
    sr_channels: Optional[Dict[int, int]] = None
    num_steps: int = 96
    ray_start: float = 2.25
    ray_end: float = 3.3
    fov: float = 18.0
    conv_clamp: Optional[float] = None
    clamp_mode: str = 'softplus'

    def sr_resolutions(self) -> List[int]:
        return [self.img_resolution // 2, self.img_resolution]

    def sr_widths(self) -> Dict[int, int]:
        if self.sr_channels is not None:
            return dict(self.sr_channels)
        r0, r1 = self.sr_resolutions()
        return {r0: min(self.channel_base // r0, self.channel_max), r1: min(self.channel_base // r1, self.channel_max)}

    def voxel_resolutions(self) -> List[int]:
        return [2 ** i for i in range(2, int(np.log2(self.plane_resolution)) + 1)]

    def voxel_width(self, res: int) -> int:
        return min(self.channel_base // res, self.channel_max)


def tiny(**overrides) -> Spec:
    base = dict(z_dim=32, c_dim=25, w_dim=32, img_resolution=64, mapping_layers=2, channel_base=256, channel_max=16,
                plane_resolution=32, plane_channels=16, render_size=8, feature_channels=8, seg_channels=5,
                decoder_hidden=32, num_steps=12)
    base.update(overrides)
    return Spec(**base)
