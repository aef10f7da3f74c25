"""CPU ORACLE for the face parser — TEST INFRASTRUCTURE ONLY (rules: header of oracle/ops.py).

Functional restatement of `BiSeNet.forward` (reference inversion/BiSeNet.py:229-256) on a state dict, eval mode:
  ConvBNReLU (BiSeNet.py:13-28)                 relu(batch_norm(conv2d(x, w, stride, padding)))   (running statistics, eps 1e-5)
  BasicBlock (resnet.py:19-47)                  relu(shortcut(x) + bn2(conv2(relu(bn1(conv1(x))))));  shortcut = bn(conv1x1 stride s) when the shape changes
  Resnet18.forward (resnet.py:70-79)            7x7 stride-2 stem, 3x3 stride-2 max pool (padding 1), four stages of two blocks -> 1/8, 1/16, 1/32 maps
  AttentionRefinementModule (BiSeNet.py:66-82)  feat * sigmoid(bn(conv1x1(mean_hw(feat))))
  ContextPath.forward (BiSeNet.py:104-125)      global context + two refinement stages, `interpolate(bilinear, align_corners=True)` between scales
  FeatureFusionModule (BiSeNet.py:178-208)      feat * sigmoid(conv2(relu(conv1(mean_hw(feat))))) + feat,  feat = convblk(cat(fsp, fcp))
  BiSeNetOutput (BiSeNet.py:36-47)              conv1x1(ConvBNReLU(x))
and of `parsing_img` / `face_parsing` (dnnlib/seg_tools.py:100-123): resize to 512 x 512, argmax, `id_remap` (:59-64), one-hot `scatter` (:92-97).
Pinned by tests/test_oracle_golden.py::test_bisenet_oracle against a reference run (tests/golden/bisenet.npz).
"""

import zlib

import torch
import torch.nn.functional as F

REMAP = (0, 1, 6, 7, 4, 5, 2, 2, 10, 11, 12, 8, 9, 15, 3, 17, 16, 18, 13, 14)          # dnnlib/seg_tools.py:59


def synthetic_state_dict(shapes):
    """A deterministic, well-conditioned state dict for {key: shape} (a BiSeNet's `state_dict()` layout): every tensor is drawn from a
    generator seeded by its KEY, so the reference model (oracle/make_golden.py) and the product / oracle (tests) get the same values whatever
    order their constructors create parameters in (the real state dict is 53 MB and is not stored)."""
    sd = {}
    for key, shape in shapes.items():
        g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7fffffff)
        shape = tuple(shape)
        if key.endswith('num_batches_tracked'):
            sd[key] = torch.zeros(shape, dtype=torch.int64)
        elif key.endswith('running_var'):
            sd[key] = torch.rand(shape, generator=g) + 0.5
        elif key.endswith('running_mean'):
            sd[key] = torch.randn(shape, generator=g) * 0.1
        elif len(shape) == 1:                                   # BatchNorm weight / bias
            sd[key] = (torch.rand(shape, generator=g) + 0.5) if key.endswith('weight') else torch.randn(shape, generator=g) * 0.1
        else:                                                   # convolution: He-scaled
            fan_in = shape[1] * shape[2] * shape[3]
            sd[key] = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
    return sd


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'], False, 0.0, 1e-5)


def _cbr(sd, p, x, stride=1, padding=1):
    return F.relu(_bn(sd, p + '.bn', F.conv2d(x, sd[p + '.conv.weight'], stride=stride, padding=padding)))


def _block(sd, p, x, stride):
    y = F.relu(_bn(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'], stride=stride, padding=1)))
    y = _bn(sd, p + '.bn2', F.conv2d(y, sd[p + '.conv2.weight'], padding=1))
    sc = x
    if p + '.downsample.0.weight' in sd:
        sc = _bn(sd, p + '.downsample.1', F.conv2d(x, sd[p + '.downsample.0.weight'], stride=stride))
    return F.relu(sc + y)


def resnet18(sd, p, x):
    x = F.relu(_bn(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for i, stride in enumerate((1, 2, 2, 2)):
        x = _block(sd, f'{p}.layer{i + 1}.0', x, stride)
        x = _block(sd, f'{p}.layer{i + 1}.1', x, 1)
        feats.append(x)
    return feats[1], feats[2], feats[3]


def _arm(sd, p, x):
    feat = _cbr(sd, p + '.conv', x)
    a = feat.mean(dim=[2, 3], keepdim=True)
    a = torch.sigmoid(_bn(sd, p + '.bn_atten', F.conv2d(a, sd[p + '.conv_atten.weight'])))
    return feat * a


def _up(x, size):
    return F.interpolate(x, tuple(size), mode='bilinear', align_corners=True)


def bisenet(sd, x):
    """-> logits [N, n_classes, H, W]"""
    sd = {k: v.float() for k, v in sd.items() if v.is_floating_point()}
    x = x.float()
    f8, f16, f32 = resnet18(sd, 'cp.resnet', x)
    avg = _cbr(sd, 'cp.conv_avg', f32.mean(dim=[2, 3], keepdim=True), padding=0)
    u32 = _arm(sd, 'cp.arm32', f32) + _up(avg, f32.shape[2:])
    u32 = _cbr(sd, 'cp.conv_head32', _up(u32, f16.shape[2:]))
    u16 = _arm(sd, 'cp.arm16', f16) + u32
    u16 = _cbr(sd, 'cp.conv_head16', _up(u16, f8.shape[2:]))
    feat = _cbr(sd, 'ffm.convblk', torch.cat([f8, u16], 1), padding=0)
    a = feat.mean(dim=[2, 3], keepdim=True)
    a = torch.sigmoid(F.conv2d(F.relu(F.conv2d(a, sd['ffm.conv1.weight'])), sd['ffm.conv2.weight']))
    feat = feat * a + feat
    out = F.conv2d(_cbr(sd, 'conv_out.conv', feat), sd['conv_out.conv_out.weight'])
    return _up(out, x.shape[2:])


def face_parsing(sd, img, classes=19, size=(512, 512)):
    """-> (one-hot labels [N, classes, *size], logits at `size`): dnnlib/seg_tools.py:120-123 + :100-117"""
    img = F.interpolate(img.float(), size=size, mode='bilinear', align_corners=True)
    logits = bisenet(sd, img)
    lab = torch.tensor(REMAP, dtype=torch.float32)[logits.argmax(1, keepdim=True)]
    return torch.zeros(img.shape[0], classes, *size).scatter_(1, lab.long(), 1), logits
