"""Face parsing for the interactive editing loop: image -> 19-class label map (second half).

Reference: `inversion/BiSeNet.py:229-281` (`BiSeNet`: ResNet-18 context path `inversion/resnet.py:57-83`, two attention refinement modules,
feature fusion, a 3x3 + 1x1 output head; the auxiliary heads `conv_out16/32` exist in the state dict and are not evaluated, :251-252) driven by
`dnnlib/seg_tools.py:100-123` (`parsing_img`, `face_parsing`: resize to 512 x 512 bilinear `align_corners=True`, argmax, `id_remap`, one-hot
`scatter`), called once per edit by Painter/run_UI.py:193-199 in front of the encoder (training/encoders.py).

Module and parameter names equal the reference's, so `segNet-20Class.pth` (dnnlib/seg_tools.py:128) loads with `load_state_dict`.
On the GPU in inference every convolution runs on the MFMA kernel of csrc/modconv.hip with its BatchNorm (eval statistics) folded into weights
and bias and the ReLU fused (lrelu with slope 0): 3x3 / 1x1 stride 1 directly, 3x3 stride 2 as explicit zero padding + the kernel's
stride-2 mode, 1x1 stride 2 on the decimated input, the 7x7 stride-2 stem as a 1x1 convolution over its unfolded 147-channel patches.
Pooling, the sigmoid gates, the residual additions and the `align_corners=True` resizes are element-wise / reduction glue and stay ATen calls
(as they are in the reference).  CPU tensors and autograd take the plain PyTorch definition.
"""

import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from training import networks

# dnnlib/seg_tools.py:59: label id of the parser's 20 classes -> the generator's 19 semantic classes
REMAP = (0, 1, 6, 7, 4, 5, 2, 2, 10, 11, 12, 8, 9, 15, 3, 17, 16, 18, 13, 14)

_fold_cache = {}


def _folded(conv, bn):
    """(weight, bias) of `bn(conv(x))` in eval mode as one convolution: w * g / sqrt(var + eps) per output channel, b = beta - mean * that.
    Cached per (parameter versions / addresses): inference only."""
    tensors = [conv.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
    stamp = tuple(networks._stamp(t) for t in tensors)
    ent = _fold_cache.get(id(conv))
    if ent is None or ent[0]() is not conv or ent[1] != stamp:
        networks._evict_dead(_fold_cache)
        w = conv.weight.detach().float()
        b = None
        if bn is not None:
            scale = bn.weight.detach().float() * (bn.running_var.float() + bn.eps).rsqrt()
            w = w * scale[:, None, None, None]
            b = (bn.bias.detach().float() - bn.running_mean.float() * scale).contiguous()
        ent = (weakref.ref(conv), stamp, w.contiguous(), b)
        _fold_cache[id(conv)] = ent
    return ent[2], ent[3]


def _on_hip(x, conv):
    return networks.use_hip_modconv and networks._inference_on_gpu(x, conv.weight) and not conv.training and networks._modconv_init()


def conv_bn_act(x, conv, bn=None, relu=False):
    """relu?(bn?(conv(x))) — one launch of the MFMA convolution kernel on the GPU in inference, the PyTorch definition otherwise."""
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    if not (_on_hip(x, conv) and (bn is None or not bn.training) and conv.bias is None and conv.groups == 1
            and (k, s, p) in ((3, 1, 1), (1, 1, 0), (3, 2, 1), (1, 2, 0), (7, 2, 3))):
        y = conv(x)
        if bn is not None:
            y = bn(y)
        return F.relu(y) if relu else y
    w, b = _folded(conv, bn)
    act, alpha = (3, 0.0) if relu else (1, 0.0)           # bias_act ids: 3 = lrelu (slope 0 = relu), 1 = linear
    mode = 0
    x = x.float()
    if (k, s) == (3, 2):
        x, mode = F.pad(x, [1, 1, 1, 1]), 1               # the kernel's stride-2 mode has no padding of its own
    elif (k, s) == (1, 2):
        x = x[:, :, ::2, ::2]
    elif k == 7:
        n, c, h, wd = x.shape
        ho, wo = (h + 2 * p - k) // s + 1, (wd + 2 * p - k) // s + 1
        x = F.unfold(x, k, padding=p, stride=s).reshape(n, c * k * k, ho, wo)          # channel order (c, ky, kx) = weight.reshape(cout, -1)
        w = w.reshape(w.shape[0], -1, 1, 1)
    return networks._modconv_plugin.modconv2d(x.contiguous(), w, None, None, None, 0.0, b, act, alpha, 1.0, -1.0, mode=mode)


class ConvBNReLU(nn.Module):
    """conv (no bias) -> BatchNorm -> ReLU (reference BiSeNet.py:13-28)."""

    def __init__(self, in_chan, out_chan, ks=3, stride=1, padding=1):
        super().__init__()
        self.conv = nn.Conv2d(in_chan, out_chan, ks, stride, padding, bias=False)
        self.bn = nn.BatchNorm2d(out_chan)
        nn.init.kaiming_normal_(self.conv.weight, a=1)

    def forward(self, x):
        return conv_bn_act(x, self.conv, self.bn, relu=True)


class BasicBlock(nn.Module):
    """ResNet basic block (reference resnet.py:19-47): relu(shortcut(x) + bn2(conv2(relu(bn1(conv1(x))))))."""

    def __init__(self, in_chan, out_chan, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_chan, out_chan, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_chan)
        self.conv2 = nn.Conv2d(out_chan, out_chan, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_chan)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if in_chan != out_chan or stride != 1:
            self.downsample = nn.Sequential(nn.Conv2d(in_chan, out_chan, 1, stride, bias=False), nn.BatchNorm2d(out_chan))

    def forward(self, x):
        y = conv_bn_act(x, self.conv1, self.bn1, relu=True)
        y = conv_bn_act(y, self.conv2, self.bn2)
        sc = x if self.downsample is None else conv_bn_act(x, self.downsample[0], self.downsample[1])
        return F.relu(sc + y)


class Resnet18(nn.Module):
    """Stem + four stages; returns the 1/8, 1/16 and 1/32 feature maps (reference resnet.py:57-83)."""

    def __init__(self, in_chan=3, out_chan=512):
        super().__init__()
        self.conv1 = nn.Conv2d(in_chan, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        widths = ((64, 64, 1), (64, 128, 2), (128, 256, 2), (256, out_chan, 2))
        for i, (ci, co, st) in enumerate(widths):
            setattr(self, f'layer{i + 1}', nn.Sequential(BasicBlock(ci, co, st), BasicBlock(co, co, 1)))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in', nonlinearity='leaky_relu')

    def forward(self, x):
        x = self.maxpool(conv_bn_act(x, self.conv1, self.bn1, relu=True))
        feat8 = self.layer2(self.layer1(x))
        feat16 = self.layer3(feat8)
        return feat8, feat16, self.layer4(feat16)


class AttentionRefinementModule(nn.Module):
    """feat * sigmoid(bn(conv1x1(global average of feat))) (reference BiSeNet.py:66-82)."""

    def __init__(self, in_chan, out_chan):
        super().__init__()
        self.conv = ConvBNReLU(in_chan, out_chan)
        self.conv_atten = nn.Conv2d(out_chan, out_chan, 1, bias=False)
        self.bn_atten = nn.BatchNorm2d(out_chan)
        self.sigmoid_atten = nn.Sigmoid()
        nn.init.kaiming_normal_(self.conv_atten.weight, a=1)

    def forward(self, x):
        feat = self.conv(x)
        atten = feat.mean(dim=[2, 3], keepdim=True)
        atten = torch.sigmoid(conv_bn_act(atten, self.conv_atten, self.bn_atten))
        return feat * atten


def _resize(x, size):
    return F.interpolate(x, size, mode='bilinear', align_corners=True)


class ContextPath(nn.Module):
    """ResNet-18 + global context + two refinement stages -> (1/8 ResNet feature, 1/8 context, 1/16 context) (reference BiSeNet.py:93-125)."""

    def __init__(self):
        super().__init__()
        self.resnet = Resnet18()
        self.arm16 = AttentionRefinementModule(256, 128)
        self.arm32 = AttentionRefinementModule(512, 128)
        self.conv_head32 = ConvBNReLU(128, 128)
        self.conv_head16 = ConvBNReLU(128, 128)
        self.conv_avg = ConvBNReLU(512, 128, ks=1, stride=1, padding=0)

    def forward(self, x):
        feat8, feat16, feat32 = self.resnet(x)
        avg = self.conv_avg(feat32.mean(dim=[2, 3], keepdim=True))
        up32 = self.arm32(feat32) + _resize(avg, feat32.shape[2:])          # a 1x1 map resized = broadcast
        up32 = self.conv_head32(_resize(up32, feat16.shape[2:]))
        up16 = self.arm16(feat16) + up32
        up16 = self.conv_head16(_resize(up16, feat8.shape[2:]))
        return feat8, up16, up32


class FeatureFusionModule(nn.Module):
    """feat * (1 + sigmoid(conv2(relu(conv1(global average of feat))))), feat = convblk(cat) (reference BiSeNet.py:178-208)."""

    def __init__(self, in_chan, out_chan):
        super().__init__()
        self.convblk = ConvBNReLU(in_chan, out_chan, ks=1, stride=1, padding=0)
        self.conv1 = nn.Conv2d(out_chan, out_chan // 4, 1, bias=False)
        self.conv2 = nn.Conv2d(out_chan // 4, out_chan, 1, bias=False)
        self.relu = nn.ReLU(inplace=True)
        self.sigmoid = nn.Sigmoid()
        for m in (self.conv1, self.conv2):
            nn.init.kaiming_normal_(m.weight, a=1)

    def forward(self, fsp, fcp):
        feat = self.convblk(torch.cat([fsp, fcp], dim=1))
        atten = feat.mean(dim=[2, 3], keepdim=True)
        atten = conv_bn_act(atten, self.conv1, relu=True)
        atten = torch.sigmoid(conv_bn_act(atten, self.conv2))
        return feat * atten + feat


class BiSeNetOutput(nn.Module):
    """3x3 conv-bn-relu + 1x1 classifier (reference BiSeNet.py:36-47)."""

    def __init__(self, in_chan, mid_chan, n_classes):
        super().__init__()
        self.conv = ConvBNReLU(in_chan, mid_chan)
        self.conv_out = nn.Conv2d(mid_chan, n_classes, 1, bias=False)
        nn.init.kaiming_normal_(self.conv_out.weight, a=1)

    def forward(self, x):
        return conv_bn_act(self.conv(x), self.conv_out)


class BiSeNet(nn.Module):
    """`net(x) -> (logits [N, n_classes, H, W], None, None)` (reference BiSeNet.py:229-256: the spatial path is the ResNet's 1/8 feature)."""

    def __init__(self, n_classes, *unused_args, **unused_kwargs):
        super().__init__()
        self.cp = ContextPath()
        self.ffm = FeatureFusionModule(256, 256)
        self.conv_out = BiSeNetOutput(256, 256, n_classes)
        self.conv_out16 = BiSeNetOutput(128, 64, n_classes)          # (training-time auxiliary heads: parameters only)
        self.conv_out32 = BiSeNetOutput(128, 64, n_classes)

    def forward(self, x):
        feat_res8, feat_cp8, _ = self.cp(x)
        out = self.conv_out(self.ffm(feat_res8, feat_cp8))
        return _resize(out, x.shape[2:]), None, None


# ---- dnnlib/seg_tools.py:59-123 ------------------------------------------------------------------------------------------------
_remap_cache = {}


def id_remap(seg, type='sof'):
    """Parser class ids -> generator class ids (float, like the reference's lookup table)."""
    key = (seg.device.type, seg.device.index)
    t = _remap_cache.get(key)
    if t is None:
        t = _remap_cache[key] = torch.tensor(REMAP, dtype=torch.float32, device=seg.device)
    return t[seg.long()]


def scatter(condition_img, classSeg=19, label_size=(512, 512)):
    """Label map [N, 1, H, W] -> one-hot [N, classSeg, *label_size] (nearest resize first when the size differs)."""
    n, _, h, w = condition_img.shape
    if (h, w) != tuple(label_size):
        condition_img = F.interpolate(condition_img, size=label_size, mode='nearest')
    out = torch.zeros(n, classSeg, *label_size, device=condition_img.device)
    return out.scatter_(1, condition_img.long(), 1)


def parsing_img(bisNet, img, argmax=True, return_mask=True, with_grad=False, remap=True):
    with torch.set_grad_enabled(bool(with_grad)):
        segmap = bisNet(img)[0]
        if argmax:
            segmap = segmap.argmax(1, keepdim=True)
        if remap:
            segmap = id_remap(segmap, 'celebahq')
    if return_mask:
        segmap = scatter(segmap)
    return img, segmap


def face_parsing(img, bisNet):
    """Image [N, 3, H, W] in [-1, 1] -> one-hot label map [N, 19, 512, 512], what the encoder's geometry tower reads."""
    img = F.interpolate(img, size=(512, 512), mode='bilinear', align_corners=True)
    return parsing_img(bisNet, img)[1]


def initFaceParsing(n_classes=20, path=None, device='cuda:0'):
    """The parser with `segNet-20Class.pth` loaded when `path` is given (dnnlib/seg_tools.py:126-135), random-init otherwise."""
    net = BiSeNet(n_classes=n_classes).to(device)
    if path is not None:
        net.load_state_dict(torch.load(path + '/segNet-20Class.pth', map_location=device))
    return net.eval().requires_grad_(False)
