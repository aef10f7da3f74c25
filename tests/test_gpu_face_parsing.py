"""The face parser of the editing loop (training/face_parsing.py; reference inversion/BiSeNet.py:229-256 + dnnlib/seg_tools.py:100-123) on the
GPU — every convolution on csrc/modconv.hip with its BatchNorm folded and the ReLU fused — against the reference-run fixture and the CPU oracle."""

import numpy as np
import pytest
import torch

from util import t

pytestmark = pytest.mark.gpu


def _net(shapes, device):
    from training import face_parsing
    from oracle import face_parsing as ofp
    net = face_parsing.BiSeNet(n_classes=20).eval().requires_grad_(False)
    sd = ofp.synthetic_state_dict(shapes)
    net.load_state_dict(sd)
    return net.to(device), sd


def test_bisenet_fixture_on_gpu(golden, gpu_device):
    from torch_utils import hip_plugin
    cases = golden('bisenet').select(fn='bisenet')
    net, _ = _net(cases[0][0]['shapes'], gpu_device)
    before = hip_plugin.CALLS.get('modconv2d', 0)
    for cfg, a in cases:
        with torch.no_grad():
            out = net(t(a['in_x'], gpu_device))[0]
        scale = float(np.abs(a['out_logits']).max())
        err = float(np.abs(out.cpu().numpy() - a['out_logits']).max())
        assert err <= 2e-5 * scale, f'logits differ from the reference run by {err / scale:.2e} of scale'
    # stem + 16 block convolutions + 3 shortcuts, 2 x (ARM conv + attention), 3 context convs, FFM (3), output head (2): every one on the HIP kernel
    assert hip_plugin.CALLS.get('modconv2d', 0) - before == 2 * 32


def test_face_parsing_full_size_against_oracle(golden, gpu_device):
    """512 x 512 (what `face_parsing` always runs at): logits vs the CPU oracle at 2e-5 of their scale; the label map and the one-hot tensor the
    encoder reads are identical wherever the oracle's two largest logits are further apart than the tolerance allows them to move."""
    from training import face_parsing
    from oracle import face_parsing as ofp
    shapes = golden('bisenet').select(fn='bisenet')[0][0]['shapes']
    net, sd = _net(shapes, gpu_device)
    g = torch.Generator().manual_seed(3)
    img = (torch.randn(1, 3, 256, 256, generator=g) * 0.5).clamp(-1, 1)
    onehot_ref, logits_ref = ofp.face_parsing(sd, img)
    with torch.no_grad():
        x512 = torch.nn.functional.interpolate(img.to(gpu_device), size=(512, 512), mode='bilinear', align_corners=True)
        logits = net(x512)[0].cpu()
        onehot = face_parsing.face_parsing(img.to(gpu_device), net).cpu()
    scale = float(logits_ref.abs().max())
    err = float((logits - logits_ref).abs().max())
    assert err <= 2e-5 * scale, f'{err / scale:.2e} of scale'
    assert onehot.shape == (1, 19, 512, 512) and onehot.dtype == torch.float32
    top2 = logits_ref.topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 2 * 2e-5 * scale                         # [1, 512, 512]
    assert decided.float().mean() > 0.99
    same = (onehot == onehot_ref).all(dim=1)
    assert bool(same[decided].all()), 'label differs where the oracle is decided'
    assert float(onehot.sum(dim=1).min()) == 1.0 and float(onehot.sum(dim=1).max()) == 1.0
