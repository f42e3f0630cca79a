"""Convolution with a fused epilogue: MIOpen runs the convolution itself (the carrier, SURVEY.md
#13), the bias / residual / ReLU that follow it are ONE in-place HIP pass (s2a_bias_act_nhwc)
instead of up to three stock elementwise kernels.  Parameter names stay ``weight`` / ``bias`` so
reference state_dicts load unchanged."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


def bias_act_(y, bias, residual=None, relu=False, out=None):
    """in place: y = act(y + bias[c] (+ residual)); y must be channels-last contiguous.
    out: a dense NHWC buffer of y's shape that receives the result instead (e.g. a level slice of a pyramid-packed tensor)"""
    _lib.require_cuda(y, bias, residual)
    B, C, H, W = y.shape
    if out is not None:
        assert out.shape == y.shape and out.dtype == y.dtype and out.permute(0, 2, 3, 1).is_contiguous()
    ok = (y.is_contiguous(memory_format=torch.channels_last) and
          C % (8 if y.dtype == torch.float16 else 4) == 0 and y.dtype in (torch.float16, torch.float32) and
          (residual is None or (residual.shape == y.shape and residual.dtype == y.dtype and
                                residual.is_contiguous(memory_format=torch.channels_last))))
    if not ok:      # odd channel count / NCHW storage: stock ops (still on the GPU)
        y = y + bias.view(1, -1, 1, 1).to(y.dtype)
        if residual is not None:
            y = y + residual
        y = F.relu(y) if relu else y
        return y if out is None else out.copy_(y)
    b = bias if bias.dtype == y.dtype and bias.is_contiguous() else bias.to(y.dtype).contiguous()
    with torch.cuda.device(y.device):
        _lib.check(_lib.lib().s2a_bias_act_nhwc_to(_lib.ptr(y), _lib.ptr(b), _lib.ptr(residual),
                                                   _lib.ptr(y if out is None else out), B * H * W, C,
                                                   _lib.dtype_code(y), int(bool(relu)), _lib.stream_ptr(y.device)))
    return y if out is None else out


def own_conv_ok(x, in_channels, out_channels, kernel_size, stride, padding, dilation, groups):
    """shapes the patch-staged MFMA convolution (s2a_conv_nhwc_f16) handles AND wins on (measured on
    MI355X, scripts/bench_ops.py --which convbb): 3x3/s1/p1 or 1x1/p0 (stride 1 or 2), f16
    channels-last, channel counts multiples of 64, enough position tiles to fill the chip"""
    import os
    if not (x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and
            x.is_contiguous(memory_format=torch.channels_last) and tuple(dilation) == (1, 1) and groups == 1 and
            (in_channels % 64 == 0 or in_channels == 32) and (out_channels % 64 == 0 or out_channels < 64) and
            x.numel() * 2 < (1 << 31)):
        return False
    if os.environ.get("S2A_NO_OWN_CONV"):
        return False
    k, st, pd = tuple(kernel_size), tuple(stride), tuple(padding)
    B, _, H, W = x.shape
    # S2A_OWN_CONV_ALWAYS=1: every shape the kernel handles, also the small grids the library wins on (the size
    # thresholds below are speed heuristics, not limits) -- a network evaluated this way runs no library convolution
    # in the trunk and is therefore bit-reproducible run to run and box to box (tests/test_net_forward.py)
    narrow = out_channels < 64 or bool(os.environ.get("S2A_OWN_CONV_ALWAYS"))
    #         ^ prediction heads (5 / 15 maps): the library's kernels for these cost 13-56 us flat
    if k == (3, 3) and st == (1, 1) and pd == (1, 1):
        return narrow or B * ((H + 7) // 8) * ((W + 15) // 16) >= 64
    if k == (3, 3) and st == (2, 2) and pd == (1, 1) and not os.environ.get("S2A_NO_OWN_CONV_S2"):
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1            # the down-sampling conv2 of a stage's first bottleneck
        return out_channels % 128 == 0 and in_channels % 64 == 0 and (narrow or B * ((Ho + 3) // 4) * ((Wo + 15) // 16) >= 128)
    if k == (1, 1) and pd == (0, 0) and st in ((1, 1), (2, 2)) and not os.environ.get("S2A_NO_OWN_CONV1"):
        Ho, Wo = (H - 1) // st[0] + 1, (W - 1) // st[1] + 1
        return narrow or B * Ho * Wo >= 64 * 128
    return False


def conv_pack_weight(weight):
    """[O,C,k,k] -> MFMA-fragment order (s2a_conv_pack_weight_f16); fewer than 64 filters are
    zero-padded to 64 (the kernel's narrowest output group)"""
    w = weight.detach().to(torch.float16).contiguous()
    if w.shape[0] < 64:
        w = torch.cat([w, w.new_zeros((64 - w.shape[0],) + tuple(w.shape[1:]))], 0).contiguous()
    if w.shape[1] == 32:            # 32 input maps (odm_cls_ls after the orientation pooling): zero-pad to one 64-chunk
        w = torch.cat([w, torch.zeros_like(w)], 1).contiguous()
    out = torch.empty_like(w)
    with torch.cuda.device(w.device):
        _lib.check(_lib.lib().s2a_conv_pack_weight_f16(_lib.ptr(w), w.shape[0], w.shape[1], w.shape[2],
                                                       _lib.ptr(out), _lib.stream_ptr(w.device)))
    return out


def conv_wino_pack_weight(weight):
    """[O,C,3,3] -> the transformed filter of the Winograd F(2,3) kernel in fragment order (s2a_conv_wino_pack_weight_f16);
    O a multiple of 64, C a multiple of 32"""
    w = weight.detach().to(torch.float16).contiguous()
    O, C = w.shape[:2]
    assert tuple(w.shape[2:]) == (3, 3) and O % 64 == 0 and C % 32 == 0
    L = _lib.lib()
    out = torch.empty((L.s2a_conv_wino_packed_elems(O, C),), dtype=torch.float16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(L.s2a_conv_wino_pack_weight_f16(_lib.ptr(w), O, C, _lib.ptr(out), _lib.stream_ptr(w.device)))
    return out


def conv_wino_f16(x, packed_wino, bias, out_channels, relu=False, pool=False):
    """relu?(conv3x3(x) + bias) on a plain channels-last tensor through the Winograd kernel (one-level pyramid)"""
    from .pyramid import PyramidLayout, conv3x3_wino
    B, C, H, W = x.shape
    assert x.dtype == torch.float16 and x.is_contiguous(memory_format=torch.channels_last)
    lay = PyramidLayout(B, [(H, W)], [1.0])
    b = None if bias is None else bias.to(torch.float16).contiguous()
    r = conv3x3_wino(lay, x.permute(0, 2, 3, 1).reshape(-1, C), packed_wino, b, out_channels, relu, pool)
    if pool:
        return (r[0].view(B, H, W, out_channels).permute(0, 3, 1, 2), r[1].view(B, H, W, out_channels // 8).permute(0, 3, 1, 2))
    return r.view(B, H, W, out_channels).permute(0, 3, 1, 2)


def conv_f16(x, packed_weight, bias, out_channels, ksize, stride=1, relu=False, residual=None, out=None):
    """relu?(conv(x) + bias (+ residual)) in ONE kernel; x f16 channels-last, packed_weight from
    conv_pack_weight; ksize 3 (pad 1, stride 1|2) or 1 (pad 0, stride 1|2)"""
    B, C, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    O_real = out_channels
    if out_channels < 64:           # narrow head: 64 physical channels, the caller gets the [:, :O] view
        assert residual is None
        out_channels = 64
    if out is None:
        out = torch.empty((B, out_channels, Ho, Wo), dtype=torch.float16, device=x.device,
                          memory_format=torch.channels_last)
    else:       # caller's buffer (e.g. a level slice of a pyramid-packed tensor): dense NHWC storage required
        assert out.shape == (B, out_channels, Ho, Wo) and out.dtype == torch.float16 and \
            out.permute(0, 2, 3, 1).is_contiguous()
    b = None if bias is None else bias.to(torch.float16).contiguous()
    if b is not None and b.numel() < out_channels:
        b = torch.cat([b, b.new_zeros(out_channels - b.numel())])
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == torch.float16 and \
            residual.is_contiguous(memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().s2a_conv_nhwc_f16(_lib.ptr(x), _lib.ptr(packed_weight), _lib.ptr(b), _lib.ptr(residual),
                                                _lib.ptr(out), B, C, H, W, out_channels, int(ksize), int(stride),
                                                int(bool(relu)), _lib.stream_ptr(x.device)))
    return out if O_real == out_channels else out[:, :O_real]


def bottleneck_tail_ok(x, conv2, conv3, residual):
    """the 64 -> 64 (3x3/s1) -> 256 (1x1) tail of a layer-1 bottleneck fits s2a_conv3x3_tail1x1_f16"""
    import os
    return (not torch.is_grad_enabled() and not os.environ.get("S2A_NO_FUSED_TAIL") and
            isinstance(conv2, FusedConv2d) and isinstance(conv3, FusedConv2d) and
            conv2.in_channels == 64 and conv2.out_channels == 64 and conv3.out_channels == 256 and
            conv2.fuse_relu and conv3.fuse_relu and conv2.bias is not None and conv3.bias is not None and
            own_conv_ok(x, 64, 64, conv2.kernel_size, conv2.stride, conv2.padding, conv2.dilation, conv2.groups) and
            tuple(conv2.kernel_size) == (3, 3) and tuple(conv2.stride) == (1, 1) and
            tuple(conv3.kernel_size) == (1, 1) and tuple(conv3.stride) == (1, 1) and
            x.shape[0] * x.shape[2] * x.shape[3] * 512 < (1 << 31) and
            (residual is None or (residual.dtype == torch.float16 and residual.shape[1] == 256 and
                                  residual.shape[2:] == x.shape[2:] and
                                  residual.is_contiguous(memory_format=torch.channels_last))))


def bottleneck_chain_ok(conv1):
    """the next block's conv1 can ride on the tail kernel: 1x1, 256 -> 64 (same stage) or 128 (first block of the
    next stage), bias, ReLU"""
    import os
    return (isinstance(conv1, FusedConv2d) and not os.environ.get("S2A_NO_TAIL_CHAIN") and
            conv1.in_channels == 256 and conv1.out_channels in (64, 128) and tuple(conv1.kernel_size) == (1, 1) and
            tuple(conv1.stride) == (1, 1) and tuple(conv1.padding) == (0, 0) and conv1.groups == 1 and
            conv1.fuse_relu and conv1.bias is not None and conv1.weight.dtype == torch.float16)


def bottleneck_tail(x, conv2, conv3, residual=None, chain=None):
    """relu(conv3(relu(conv2(x))) + residual) in ONE kernel (models/backbone.py:72-83, BN folded): the 64-map
    intermediate stays in LDS; bit-identical to the two separate launches.  chain = the NEXT bottleneck's conv1
    (bottleneck_chain_ok): its output relu(conv1(out)) is produced by the same launch -> returns (out, next_conv1_out)"""
    B, C, H, W = x.shape
    out = torch.empty((B, 256, H, W), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    w2, b2, _ = conv2.packed_args()
    w3, b3, _ = conv3.packed_args()
    wc = bc = nxt = None
    if chain is not None:
        wc, bc, _ = chain.packed_args()
        nxt = torch.empty((B, chain.out_channels, H, W), dtype=torch.float16, device=x.device,
                          memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().s2a_conv3x3_tail1x1_f16(_lib.ptr(x), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(w3), _lib.ptr(b3),
                                                      _lib.ptr(residual), _lib.ptr(out), _lib.ptr(wc), _lib.ptr(bc),
                                                      _lib.ptr(nxt), 0 if chain is None else chain.out_channels,
                                                      B, 64, 64, 256, H, W, _lib.stream_ptr(x.device)))
    return out if chain is None else (out, nxt)


def conv1x1_add_up2(x, packed_weight, bias, coarse, out_channels):
    """FPN top-down step (models/neck.py:67-79) in one launch:
    conv1x1(x) + bias + nearest-2x-upsample(coarse); x[B,C,H,W], coarse[B,O,H/2,W/2] f16 channels-last"""
    B, C, H, W = x.shape
    assert coarse.shape == (B, out_channels, H // 2, W // 2) and H % 2 == 0 and W % 2 == 0
    assert coarse.dtype == torch.float16 and coarse.is_contiguous(memory_format=torch.channels_last)
    out = torch.empty((B, out_channels, H, W), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().s2a_conv1x1_add_up2_f16(_lib.ptr(x), _lib.ptr(packed_weight), _lib.ptr(bias), _lib.ptr(coarse),
                                                      _lib.ptr(out), B, C, H, W, out_channels, _lib.stream_ptr(x.device)))
    return out


class PackedWeightCache:
    """inference-time cache of a conv_pack_weight() result, invalidated when the tensor changes"""

    def __init__(self):
        self.key, self.val = None, None
        self.bkey, self.bval = None, None
        self.wkey, self.wval = None, None

    def get_wino(self, w):
        """the Winograd-transformed filter (conv_wino_pack_weight) of a [O,C,3,3] weight"""
        key = (w._version, w.data_ptr(), w.device)
        if self.wkey != key:
            self.wkey, self.wval = key, conv_wino_pack_weight(w)
        return self.wval

    def get(self, w):
        key = (w._version, w.data_ptr(), w.device)
        if self.key != key:
            self.key, self.val = key, conv_pack_weight(w)
        return self.val

    def get_bias(self, b, width):
        """f16 bias zero-padded to the physical channel count"""
        if b is None:
            return None
        key = (b._version, b.data_ptr(), b.device, width)
        if self.bkey != key:
            v = b.detach().to(torch.float16)
            if v.numel() < width:
                v = torch.cat([v, v.new_zeros(width - v.numel())])
            self.bkey, self.bval = key, v.contiguous()
        return self.bval


class FusedConv2d(nn.Conv2d):
    """nn.Conv2d + (bias, optional residual, optional ReLU) epilogue in one pass"""

    def __init__(self, *args, relu=False, **kw):
        super().__init__(*args, **kw)
        self.fuse_relu = relu

    @classmethod
    def from_conv(cls, conv, relu=False):
        m = cls(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                conv.dilation, conv.groups, bias=conv.bias is not None, relu=relu)
        m = m.to(conv.weight.device, conv.weight.dtype)
        m.weight = conv.weight
        m.bias = conv.bias
        return m

    def packed_args(self):
        """(fragment-order filter, f16 bias, physical out channels) for the pyramid-packed launches"""
        if not hasattr(self, "_packed"):
            self._packed = PackedWeightCache()
        width = max(64, self.out_channels)
        return self._packed.get(self.weight), self._packed.get_bias(self.bias, width), width

    def wino_ok(self):
        """a layer the Winograd kernel serves: 3x3 / stride 1 / pad 1, O a multiple of 64, C a multiple of 32"""
        return (tuple(self.kernel_size) == (3, 3) and tuple(self.stride) == (1, 1) and tuple(self.padding) == (1, 1) and
                tuple(self.dilation) == (1, 1) and self.groups == 1 and self.out_channels % 64 == 0 and
                self.in_channels % 32 == 0)

    def packed_args_wino(self):
        """(transformed filter, f16 bias, out channels) for s2a_conv3x3_wino_pyramid_f16"""
        if not hasattr(self, "_packed"):
            self._packed = PackedWeightCache()
        return self._packed.get_wino(self.weight), self._packed.get_bias(self.bias, self.out_channels), self.out_channels

    def forward(self, x, residual=None, out=None):
        """out: dense NHWC buffer for the result (library path only; the own kernel is called with out= directly)"""
        if out is not None:
            assert x.is_cuda and self.bias is not None and not torch.is_grad_enabled() and not own_conv_ok(
                x, self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding, self.dilation, self.groups)
            y = F.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation, self.groups)
            return bias_act_(y, self.bias, residual, self.fuse_relu, out=out)
        if not torch.is_grad_enabled() and own_conv_ok(
                x, self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding,
                self.dilation, self.groups) and (residual is None or (
                    residual.dtype == torch.float16 and residual.is_contiguous(memory_format=torch.channels_last))):
            if not hasattr(self, "_packed"):
                self._packed = PackedWeightCache()
            return conv_f16(x, self._packed.get(self.weight), self._packed.get_bias(self.bias, max(64, self.out_channels)),
                            self.out_channels, self.kernel_size[0], self.stride[0], self.fuse_relu, residual)
        if (not x.is_cuda) or self.bias is None:
            y = super().forward(x)
            if residual is not None:
                y = y + residual
            return F.relu(y) if self.fuse_relu else y
        y = F.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation, self.groups)
        return bias_act_(y, self.bias, residual, self.fuse_relu)


def stem_pack_weight(weight):
    """[64,3,7,7] -> the fragment order of the fused stem kernel (s2a_stem_pack_weight_f16)"""
    w = weight.detach().to(torch.float16).contiguous()
    assert tuple(w.shape) == (64, 3, 7, 7)
    L = _lib.lib()
    out = torch.empty((L.s2a_stem_packed_elems(),), dtype=torch.float16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(L.s2a_stem_pack_weight_f16(_lib.ptr(w), _lib.ptr(out), _lib.stream_ptr(w.device)))
    return out


def stem_u8(imgs_u8, packed_weight, bias, divisor=255.0):
    """uint8 image [B,3,H,W] (channels-last storage) -> relu(conv7x7/2(img / divisor) + bias) -> maxpool 3x3/2:
    [B,64,H/4,W/4] f16 channels-last, one kernel (s2a_stem_u8_f16)"""
    _lib.require_cuda(imgs_u8)
    B, C, H, W = imgs_u8.shape
    assert C == 3 and imgs_u8.dtype == torch.uint8 and imgs_u8.permute(0, 2, 3, 1).is_contiguous()
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    out = torch.empty((B, 64, Hp, Wp), dtype=torch.float16, device=imgs_u8.device, memory_format=torch.channels_last)
    b = None if bias is None else bias.to(torch.float16).contiguous()
    with torch.cuda.device(imgs_u8.device):
        _lib.check(_lib.lib().s2a_stem_u8_f16(_lib.ptr(imgs_u8), _lib.ptr(packed_weight), _lib.ptr(b), _lib.ptr(out),
                                              B, H, W, float(divisor), _lib.stream_ptr(imgs_u8.device)))
    return out
