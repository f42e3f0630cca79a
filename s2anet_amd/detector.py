"""Carrier network for the end-to-end configs: R-50 backbone (C3-C5) + FPN (P3-P7) + S2ANetHead.

The backbone and neck carry no custom arithmetic: their layers are ``torch.nn`` modules that hold
the parameters, and at f16 inference every 1x1 / 3x3 convolution of them runs on this package's own ``k_conv_f16``
(BN folded, bias / residual / ReLU fused; `fused.py`) - only FPN's two stride-2 extra levels go to the library.
Structure follows
models/backbone.py:37-175,283-354 and models/neck.py:5-96 (same module/parameter names, so a
reference ``state_dict`` loads), but the constructor is OFFLINE: no torchvision download
(backbone.py:241-255); weights are seeded random (kaiming for ResNet backbone.py:134-140,
xavier-uniform for FPN neck.py:58-62, N(0,0.01) for the head).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .fused import FusedConv2d, bottleneck_chain_ok, bottleneck_tail, bottleneck_tail_ok
from .head import S2ANetHead


class BottleNeck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        return self.forward_chain(x)[0]

    def forward_chain(self, x, pre=None, nxt=None):
        """-> (block output, the NEXT block's conv1 output or None).  pre = this block's conv1 output when the previous
        block's tail kernel already produced it; nxt = the next block (run_blocks)."""
        residual = x
        if isinstance(self.conv3, FusedConv2d):           # BN folded, epilogues fused (inference)
            out = pre if pre is not None else self.conv1(x)
            if self.downsample is not None:
                residual = self.downsample(x)
            if bottleneck_tail_ok(out, self.conv2, self.conv3, residual):   # conv2 + conv3 + residual: one launch
                if isinstance(nxt, BottleNeck) and bottleneck_chain_ok(nxt.conv1):
                    return bottleneck_tail(out, self.conv2, self.conv3, residual, nxt.conv1)
                return bottleneck_tail(out, self.conv2, self.conv3, residual), None
            return self.conv3(self.conv2(out), residual), None  # relu(conv3 + bias + residual) in one pass
        return self._forward_plain(x), None

    def _forward_plain(self, x):
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        out += residual
        return self.relu(out)


class DetectorBackbone(nn.Module):
    """ResNet-50 trunk exposing C3, C4, C5 (out_indices=(2,3,4), backbone.py:283-354)"""

    def __init__(self, layers=(3, 4, 6, 3), out_indices=(2, 3, 4)):
        super().__init__()
        self.inplanes = 64
        conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        bn1 = nn.BatchNorm2d(64)
        layer1 = self._make_layer(64, layers[0])
        layer2 = self._make_layer(128, layers[1], stride=2)
        layer3 = self._make_layer(256, layers[2], stride=2)
        layer4 = self._make_layer(512, layers[3], stride=2)
        self.backbone = nn.Sequential(
            nn.Sequential(conv1, bn1, nn.ReLU(inplace=True)),
            nn.Sequential(nn.MaxPool2d(kernel_size=3, stride=2, padding=1, ceil_mode=False), layer1),
            layer2, layer3, layer4)
        self.out_indices = out_indices
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [BottleNeck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        layers += [BottleNeck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    @staticmethod
    def run_blocks(seq, x, pre=None, after=None, with_pre=False):
        """a stage's bottlenecks one after the other; a block's tail kernel may hand the next block its conv1 output.
        pre = conv1 output for the first block (from the previous stage's last tail), after = the block that follows
        the stage; with_pre -> (x, conv1 output for `after` or None)"""
        for i, blk in enumerate(seq):
            if not isinstance(blk, BottleNeck):
                x, pre = blk(x), None
                continue
            x, pre = blk.forward_chain(x, pre, seq[i + 1] if i + 1 < len(seq) else after)
        return (x, pre) if with_pre else x

    def _stages(self, x):
        """layer1 .. layer4 on the pooled stem output"""
        outs, pre = [], None
        for i in range(1, len(self.backbone)):
            seq = self.backbone[i][1] if i == 1 else self.backbone[i]
            after = self.backbone[i + 1][0] if i + 1 < len(self.backbone) else None
            x, pre = self.run_blocks(seq, x, pre, after, with_pre=True)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def forward(self, x):
        assert 0 not in self.out_indices
        return self._stages(self.backbone[1][0](self.backbone[0](x)))

    def stem_fusable(self, imgs_u8):
        """uint8 channels-last batch + BN-folded stem: /255, conv1, ReLU and the max-pool run as one kernel"""
        conv, pool = self.backbone[0][0], self.backbone[1][0]
        return (isinstance(conv, FusedConv2d) and conv.fuse_relu and conv.weight.dtype == torch.float16 and
                tuple(conv.weight.shape) == (64, 3, 7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and
                isinstance(pool, nn.MaxPool2d) and pool.kernel_size == 3 and pool.stride == 2 and pool.padding == 1 and
                not pool.ceil_mode and imgs_u8.is_cuda and imgs_u8.dtype == torch.uint8 and imgs_u8.shape[1] == 3 and
                imgs_u8.shape[3] % 4 == 0 and min(imgs_u8.shape[2:]) >= 7 and
                imgs_u8.permute(0, 2, 3, 1).is_contiguous() and not torch.is_grad_enabled() and
                not os.environ.get("S2A_NO_FUSED_STEM"))

    def forward_u8(self, imgs_u8, divisor=255.0):
        """forward() on the raw uint8 batch with the fused stem (s2a_stem_u8_f16)"""
        from .fused import PackedWeightCache, stem_pack_weight, stem_u8
        conv = self.backbone[0][0]
        w = conv.weight
        key = (w._version, w.data_ptr(), w.device)
        if getattr(self, "_stem_key", None) != key:
            self._stem_key, self._stem_w = key, stem_pack_weight(w)
            self._stem_b = None if conv.bias is None else conv.bias.detach().to(torch.float16).contiguous()
        return self._stages(stem_u8(imgs_u8, self._stem_w, self._stem_b, divisor))


class FPN(nn.Module):
    def __init__(self, in_channels=(512, 1024, 2048), out_channels=256, num_outs=5):
        super().__init__()
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.lateral_convs = nn.ModuleList(nn.Conv2d(c, out_channels, 1) for c in in_channels)
        self.fpn_convs = nn.ModuleList(nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels)
        for i in range(num_outs - self.num_ins):
            cin = in_channels[-1] if i == 0 else out_channels
            self.fpn_convs.append(nn.Conv2d(cin, out_channels, 3, stride=2, padding=1))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    def forward_packed(self, inputs, layout):
        """same as forward(), but the five outputs land back to back in ONE pyramid-packed buffer
        [sum B*H*W, 256] (s2anet_amd/pyramid.py) so that the head can run each layer once for all levels"""
        from .fused import conv_f16, conv1x1_add_up2, own_conv_ok
        lat = [None] * self.num_ins
        lat[-1] = self.lateral_convs[-1](inputs[-1])
        for i in range(self.num_ins - 1, 0, -1):           # top-down: lateral conv + up-sampled coarser level, one launch
            l, x = self.lateral_convs[i - 1], inputs[i - 1]
            if (hasattr(l, "packed_args") and x.shape[2] == 2 * lat[i].shape[2] and x.shape[3] == 2 * lat[i].shape[3] and
                    lat[i].is_contiguous(memory_format=torch.channels_last) and l.out_channels % 64 == 0 and
                    own_conv_ok(x, l.in_channels, l.out_channels, l.kernel_size, l.stride, l.padding, l.dilation, l.groups)):
                w, b, o = l.packed_args()
                lat[i - 1] = conv1x1_add_up2(x, w, b, lat[i], o)
            else:
                lat[i - 1] = l(x) + F.interpolate(lat[i], scale_factor=2, mode="nearest")
        buf = layout.new(self.fpn_convs[0].out_channels, inputs[0].device)
        for i in range(self.num_outs):
            conv = self.fpn_convs[i]
            src = lat[i] if i < self.num_ins else (inputs[-1] if i == self.num_ins else prev)
            dst = layout.level(buf, i)                          # [B,C,H,W] channels-last view of the slice
            if i < self.num_ins and hasattr(conv, "packed_args"):
                w, b, o = conv.packed_args()
                prev = conv_f16(src, w, b, o, 3, 1, False, out=dst)
            elif isinstance(conv, FusedConv2d) and conv.bias is not None and src.is_contiguous(memory_format=torch.channels_last) \
                    and not own_conv_ok(src, conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride,
                                        conv.padding, conv.dilation, conv.groups):
                prev = conv(src, out=dst)       # library convolution, its bias pass writes the pyramid slice (no copy)
            else:
                prev = conv(src)
                dst.copy_(prev)
        return buf

    def forward(self, inputs):
        lat = [l(inputs[i]) for i, l in enumerate(self.lateral_convs)]
        for i in range(self.num_ins - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], scale_factor=2, mode="nearest")
        outs = [self.fpn_convs[i](lat[i]) for i in range(self.num_ins)]
        for i in range(self.num_outs - self.num_ins):
            outs.append(self.fpn_convs[self.num_ins + i](inputs[-1] if i == 0 else outs[-1]))
        return tuple(outs)


class S2ANet(nn.Module):
    """models/detector.py:9-37.  forward(imgs, post_process) -> dict like the reference."""

    def __init__(self, num_classes=15, **head_kw):
        super().__init__()
        self.stride = (8, 16, 32, 64, 128)
        self.backbone = DetectorBackbone()
        self.neck = FPN(num_outs=len(self.stride))
        self.head = S2ANetHead(num_classes=num_classes, featmap_strides=self.stride, **head_kw)

    def forward(self, imgs, post_process=False):
        return self.head(self.neck(self.backbone(imgs)), post_process=post_process)

    def features_to_pred(self, imgs, backbone_out=None, **pyramid_kw):
        """pyramid_kw: ``anchors`` / ``trace`` of S2ANetHead.forward_pyramid (pyramid-packed path only)"""
        c = self.backbone(imgs) if backbone_out is None else backbone_out
        sizes = [tuple(c[0].shape[2:])]
        while len(sizes) < len(self.stride):                    # stride-2 3x3/pad-1 convs: ceil(n/2)
            sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
        if self.head.pyramid_ok(c[0]) and hasattr(self.neck.fpn_convs[0], "packed_args") and \
                min(sizes[-1]) >= 3 and not os.environ.get("S2A_NO_PYRAMID"):
            from .pyramid import PyramidLayout
            B = imgs.shape[0]
            key = (B, tuple(sizes))
            if getattr(self, "_layout_key", None) != key:
                self._layout_key, self._layout = key, PyramidLayout(B, sizes, self.stride)
            if pyramid_kw.get("trace") is not None:
                pyramid_kw["trace"].update(C=c, layout=self._layout)
            return self.head.forward_pyramid(self._layout, self.neck.forward_packed(c, self._layout), **pyramid_kw)
        assert not pyramid_kw, "anchors= / trace= need the pyramid-packed path"
        feats = self.neck(c)
        per_level = [self.head.forward_single(f, s) for f, s in zip(feats, self.stride)]
        return tuple(map(list, zip(*per_level)))

    @torch.no_grad()
    def detect(self, imgs_u8, max_candidates=None, return_overflow=False, **nms_kw):
        """device-resident uint8 batch [B,3,H,W] -> (dets[B,2000,6], labels[B,2000], counts[B]).
        /255 normalisation as val.py:246-247; no host synchronisation anywhere.
        max_candidates caps the (box, class) rows the batch's NMS considers (static shapes); the reference never
        drops one, so with return_overflow=True a fourth result int64[2] = [candidates found, candidates dropped]
        comes back on the device (dropped must be 0 for reference-equal results).
        nms_kw: ``dropped_total`` (device int64 accumulator) and ``return_wire`` (append the all-gather wire buffer
        float32 [B, 2000*7+1] that ``dets`` is a view of) -- see rotated.batched_multiclass_nms_rotated."""
        if self.backbone.stem_fusable(imgs_u8):
            return self.head.get_bboxes_batched(
                self.features_to_pred(imgs_u8, self.backbone.forward_u8(imgs_u8, 255.0)), max_candidates, return_overflow,
                **nms_kw)
        dt = next(self.parameters()).dtype
        x = imgs_u8.to(dt).div_(255.0)
        if imgs_u8.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        return self.head.get_bboxes_batched(self.features_to_pred(x), max_candidates, return_overflow, **nms_kw)


def load_reference_checkpoint(model, weights, map_location="cpu"):
    """val.py:153-183: a ``.pth`` file (or an already loaded dict) with a ``"state_dict"`` entry is
    matched to the model's own entries BY POSITION (the official-code checkpoints use other key
    names; ``intersect_dicts`` zips the two ordered dicts and asserts equal length), then loaded
    strictly.  A dict with a ``"model"`` entry holds a module (or its state_dict): loaded by name.
    Call before ``fold_batchnorm`` / ``fuse_epilogues`` (they keep values, not the BN entries)."""
    ckpt = torch.load(weights, map_location=map_location, weights_only=False) \
        if isinstance(weights, (str, os.PathLike)) else weights
    own = model.state_dict()
    if "state_dict" in ckpt:
        src = ckpt["state_dict"]
        if len(src) != len(own):
            raise AssertionError(f"checkpoint has {len(src)} entries, the model {len(own)}")
        mapped = {}
        for (k1, v1), (k2, v2) in zip(own.items(), src.items()):
            if tuple(v1.shape) != tuple(v2.shape):
                raise RuntimeError(f"size mismatch for {k1} <- {k2}: {tuple(v2.shape)} vs {tuple(v1.shape)}")
            mapped[k1] = v2
        model.load_state_dict(mapped, strict=True)
    elif "model" in ckpt:
        src = ckpt["model"]
        model.load_state_dict(src.state_dict() if hasattr(src, "state_dict") else src, strict=True)
    else:
        raise KeyError("checkpoint holds neither 'state_dict' nor 'model' (val.py:155,182)")
    return model


def fold_batchnorm(model):
    """inference-time conv+BN folding (BN in eval mode is an affine map): fewer passes over HBM"""
    def fold(conv, bn):
        w = conv.weight
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                          conv.dilation, conv.groups, bias=True).to(w.device, w.dtype)
        fused.weight.data = w * scale.view(-1, 1, 1, 1)
        b = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
        fused.bias.data = (b - bn.running_mean) * scale + bn.bias
        return fused

    for mod in model.modules():
        if isinstance(mod, BottleNeck):
            mod.conv1, mod.bn1 = FusedConv2d.from_conv(fold(mod.conv1, mod.bn1), relu=True), nn.Identity()
            mod.conv2, mod.bn2 = FusedConv2d.from_conv(fold(mod.conv2, mod.bn2), relu=True), nn.Identity()
            mod.conv3, mod.bn3 = FusedConv2d.from_conv(fold(mod.conv3, mod.bn3), relu=True), nn.Identity()
            if mod.downsample is not None:
                mod.downsample = nn.Sequential(FusedConv2d.from_conv(fold(mod.downsample[0], mod.downsample[1])))
    stem = model.backbone.backbone[0]
    model.backbone.backbone[0] = nn.Sequential(FusedConv2d.from_conv(fold(stem[0], stem[1]), relu=True), nn.Identity())
    return model


def fuse_epilogues(model):
    """swap every remaining biased nn.Conv2d (FPN, head towers, prediction heads) for FusedConv2d;
    conv+ReLU pairs inside nn.Sequential collapse into one module (names/params unchanged)"""
    from .orn import ORConv2d
    for parent in list(model.modules()):
        if isinstance(parent, nn.Sequential) and len(parent) == 2 and type(parent[0]) is nn.Conv2d \
                and isinstance(parent[1], nn.ReLU) and parent[0].bias is not None:
            parent[0] = FusedConv2d.from_conv(parent[0], relu=True)
            parent[1] = nn.Identity()
    for parent in list(model.modules()):
        for name, child in list(parent.named_children()):
            if type(child) is nn.Conv2d and child.bias is not None and not isinstance(child, ORConv2d):
                setattr(parent, name, FusedConv2d.from_conv(child, relu=False))
    return model


def build_synthetic_detector(num_classes=15, seed=1234, dtype=torch.float16, device="cuda",
                             channels_last=True, fold_bn=True, **head_kw):
    """random-init S2ANet of the reference architecture (no network access, no checkpoint).
    The last BN of every bottleneck gets gamma = 0.25 so that activations of the untrained
    50-layer trunk stay inside fp16 range (a trained network has learned scales instead)."""
    torch.manual_seed(seed)
    m = S2ANet(num_classes=num_classes, **head_kw)
    for mod in m.modules():
        if isinstance(mod, BottleNeck):
            mod.bn3.weight.data.fill_(0.25)
    m.eval()
    if fold_bn:
        fold_batchnorm(m)
        fuse_epilogues(m)
    m = m.to(device=device, dtype=dtype)
    if channels_last:
        # 4-D conv filters only (ORConv2d keeps its 5-D filter bank; its cached ARF expansion is
        # converted in ORConv2d.rotate_arf)
        for mod in m.modules():
            if isinstance(mod, nn.Conv2d) and mod.weight.dim() == 4 and mod.weight.shape[1] >= 8:
                mod.weight.data = mod.weight.data.contiguous(memory_format=torch.channels_last)
        m.head.or_conv.channels_last = True
    return m
