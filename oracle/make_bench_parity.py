"""TEST / BENCH-CHECK INFRASTRUCTURE ONLY (rules: header of oracle/ops.py).

Writes tests/golden/bench_parity.npz: what the CPU ORACLE (fp32 torch-CPU restatement, oracle/generator.py + fast_ops.py,
itself pinned to reference-run golden vectors by tests/test_oracle_golden.py) produces for the exact inputs `bench.py` feeds
its parity step — full-size random-init ide3d-ffhq-64-512 generator (torch.manual_seed(0)), seeds 0..3, bench.py's four yaws,
stratified-jitter draws of torch.Generator().manual_seed(PARITY_JITTER_SEED).  `bench.py` renders the same inputs through the
benchmarked hipGraph after its timed region and prints `parity_ok` (fixture only — the timed path never touches `oracle/`).

The fixture stores sub-sampled outputs (the full fp32 frames would be 90 MB): the raw 64x64 render of every image, every 8th
pixel of the 512^2 RGB image and every 16th pixel of the 19 seg-logit maps, plus the oracle's own scale of each tensor.

    python oracle/make_bench_parity.py            # ~1 min on 8 cores
"""

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, 'ide-3d_amd'), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PARITY_SEEDS = (0, 1, 2, 3)
PARITY_YAWS = (-0.5, 0.0, 0.5, 0.25)
PARITY_JITTER_SEED = 11
IMG_STRIDE, SEG_STRIDE = 8, 16


def parity_inputs():
    """(z [4, 512] float64, cams [4, 25], cond [4, 25], jitter [4, 4096, 96]) — shared by this script, bench.py and the tests."""
    from training import triplane
    z = torch.from_numpy(np.stack([np.random.RandomState(s).randn(512) for s in PARITY_SEEDS]))
    cams = torch.cat([triplane.camera_label(y) for y in PARITY_YAWS])
    cond = triplane.conditioning_label().repeat(len(PARITY_SEEDS), 1)
    jit = torch.rand(len(PARITY_SEEDS), 4096, 96, generator=torch.Generator().manual_seed(PARITY_JITTER_SEED))
    return z, cams, cond, jit


def subsample(img, seg, raw):
    return dict(img=img[:, :, ::IMG_STRIDE, ::IMG_STRIDE].contiguous(), seg=seg[:, :, ::SEG_STRIDE, ::SEG_STRIDE].contiguous(), raw=raw.contiguous())


def main():
    from oracle import fast_ops, generator as ogen, spec as ospec
    from training import triplane
    torch.manual_seed(0)
    G = triplane.TriPlaneGenerator().eval()
    sd = {k: v.detach() for k, v in G.state_dict().items()}
    sp = ospec.Spec()
    z, cams, cond, jit = parity_inputs()
    parts = []
    for i in range(len(PARITY_SEEDS)):
        ws = ogen.mapping(sd, sp, z[i:i + 1], cond[i:i + 1], ops=fast_ops)
        ref = ogen.synthesis(sd, sp, ws, cams[i:i + 1], jitter=jit[i:i + 1], ops=fast_ops)
        parts.append((ref['image'], ref['image_seg'], ref['image_raw'], ws))
        print(f'seed {PARITY_SEEDS[i]}: image scale {float(ref["image"].abs().max()):.3f}, seg scale {float(ref["image_seg"].abs().max()):.3f}')
    img, seg, raw, ws = (torch.cat([p[k] for p in parts]) for k in range(4))
    sub = subsample(img, seg, raw)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'bench_parity.npz'),
                        img=sub['img'].numpy(), seg=sub['seg'].numpy(), raw=sub['raw'].numpy(), ws0=ws[:, 0].numpy(),
                        scale_img=np.float32(img.abs().max()), scale_seg=np.float32(seg.abs().max()), scale_raw=np.float32(raw.abs().max()),
                        torch_version=np.bytes_(torch.__version__.encode()))
    print('wrote tests/golden/bench_parity.npz')


if __name__ == '__main__':
    main()
