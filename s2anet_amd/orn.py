"""ORN host side: active rotating filters + rotation-invariant pooling.

Mirrors models/orn:
    orn_cuda.arf_forward(weight, indices)                     models/orn/src/vision.cpp:7-12
    active_rotating_filter = _ActiveRotatingFilter.apply       functions/active_rotating_filter.py:11-34
    ORConv2d(in, out, kernel_size, arf_config, ...)            modules/ORConv.py:12-101
    RotationInvariantPooling(nInputPlane, nOrientation=8)      functions/rotation_invariant_pooling.py:6-27
State-dict entries of ORConv2d: ``weight [O,I,nOri,k,k]``, ``bias [O*nRot]``, ``indices`` (uint8).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.utils import _pair

from . import _lib

# 3x3 tap permutation of a rotation by k*45 degrees (1-based), ORConv.py:53-62
_ROT3 = {0: (1, 2, 3, 4, 5, 6, 7, 8, 9), 45: (2, 3, 6, 1, 5, 9, 4, 7, 8),
         90: (3, 6, 9, 2, 5, 8, 1, 4, 7), 135: (6, 9, 8, 3, 5, 7, 2, 1, 4),
         180: (9, 8, 7, 6, 5, 4, 3, 2, 1), 225: (8, 7, 4, 9, 5, 1, 6, 3, 2),
         270: (7, 4, 1, 8, 5, 2, 9, 6, 3), 315: (4, 1, 2, 7, 5, 3, 8, 9, 6)}
_ROT1 = {a: (1,) for a in range(0, 360, 45)}


def arf_forward(weight, indices):
    """orn_cuda.arf_forward: weight[O,I,nOri,kH,kW], indices uint8[nOri,kH,kW,nRot]
    -> [O*nRot, I*nOri, kH, kW] (new tensor)."""
    _lib.require_cuda(weight, indices)
    if weight.dim() != 5:
        raise RuntimeError("only supports a batch of ARFs.")   # ARF_forward_cuda:81
    w = weight.contiguous()
    idx = indices.contiguous()
    if idx.dtype != torch.uint8:
        idx = idx.byte()
    O, I, nOri, kH, kW = w.shape
    nRot = idx.shape[3]
    out = torch.empty((O * nRot, I * nOri, kH, kW), dtype=w.dtype, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.lib().s2a_arf_forward(_lib.ptr(w), _lib.ptr(idx), O, I, nOri, kH, kW, nRot,
                                              _lib.DTYPE_F64 if w.dtype == torch.float64 else _lib.dtype_code(w), _lib.ptr(out),
                                              _lib.stream_ptr(w.device)))
    return out


def arf_backward(indices, grad_output):
    """orn_cuda.arf_backward: indices uint8[nOri,kH,kW,nRot], gradOutput[O*nRot, I*nOri, kH, kW]
    -> gradInput[O, I, nOri, kH, kW] (new tensor; float32 or float64)"""
    _lib.require_cuda(indices, grad_output)
    idx = indices.contiguous()
    if idx.dtype != torch.uint8:
        idx = idx.byte()
    g = grad_output.contiguous()
    if g.dtype not in (torch.float32, torch.float64):     # the reference's dispatch (ActiveRotatingFilter_cuda.cu:149)
        raise RuntimeError("arf_backward: float32 / float64 gradients only")
    nOri, kH, kW, nRot = idx.shape
    O, I = g.shape[0] // nRot, g.shape[1] // nOri
    out = torch.empty((O, I, nOri, kH, kW), dtype=g.dtype, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(_lib.lib().s2a_arf_backward(_lib.ptr(idx), _lib.ptr(g), O, I, nOri, kH, kW, nRot,
                                               _lib.DTYPE_F64 if g.dtype == torch.float64 else _lib.dtype_code(g),
                                               _lib.ptr(out), _lib.stream_ptr(g.device)))
    return out


def rie_forward(feature, n_orientation):
    """orn_cuda.rie_forward(feature[B,C,1,1], nOrientation) -> (mainDirection uint8[B,C/nOri], aligned[B,C,1,1])
    (models/orn/src/vision.cpp:10, RotationInvariantEncoding.h:11-22); float32"""
    _lib.require_cuda(feature)
    if feature.dim() != 4:
        raise RuntimeError("only supports a batch of RIEs.")                       # AT_ASSERTM of the reference
    if feature.shape[2] != 1 or feature.shape[3] != 1:
        raise RuntimeError("mH x mW should be 1x1.")
    if feature.dtype != torch.float32:
        raise RuntimeError("rie_forward: float32 features only")
    f = feature.contiguous()
    B, C = f.shape[:2]
    n = int(n_orientation)
    direction = torch.empty((B, C // n), dtype=torch.uint8, device=f.device)
    aligned = torch.zeros_like(f)
    with torch.cuda.device(f.device):
        _lib.check(_lib.lib().s2a_rie_forward(_lib.ptr(f), B, C, n, _lib.dtype_code(f), _lib.ptr(direction),
                                              _lib.ptr(aligned), _lib.stream_ptr(f.device)))
    return direction, aligned


def rie_backward(main_direction, grad_output, n_orientation):
    """orn_cuda.rie_backward(mainDirection uint8[B,F], gradOutput[B,F*nOri,1,1], nOrientation) -> gradInput"""
    _lib.require_cuda(main_direction, grad_output)
    if grad_output.dtype != torch.float32:
        raise RuntimeError("rie_backward: float32 gradients only")
    d = main_direction.contiguous()
    if d.dtype != torch.uint8:
        d = d.byte()
    g = grad_output.contiguous()
    B, F = d.shape
    out = torch.zeros_like(g)
    with torch.cuda.device(g.device):
        _lib.check(_lib.lib().s2a_rie_backward(_lib.ptr(d), _lib.ptr(g), B, F, int(n_orientation), _lib.dtype_code(g),
                                               _lib.ptr(out), _lib.stream_ptr(g.device)))
    return out


class _RotationInvariantEncoding(torch.autograd.Function):
    """models/orn/functions/rotation_invariant_encoding.py:11-38"""

    @staticmethod
    def forward(ctx, input, nOrientation, return_direction=False):
        ctx.nOrientation = nOrientation
        ctx.return_direction = return_direction
        mainDirection, output = rie_forward(input, nOrientation)
        if return_direction:
            ctx.save_for_backward(input, mainDirection)
            ctx.mark_non_differentiable(mainDirection)
            return output, mainDirection
        ctx.save_for_backward(input)
        ctx.mainDirection = mainDirection
        return output

    @staticmethod
    def backward(ctx, grad_output, *unused):
        if ctx.return_direction:
            _, mainDirection = ctx.saved_tensors
        else:
            mainDirection = ctx.mainDirection
        return rie_backward(mainDirection, grad_output.contiguous(), ctx.nOrientation), None, None


class RotationInvariantEncoding(torch.nn.Module):
    """models/orn/functions/rotation_invariant_encoding.py:41-49"""

    def __init__(self, nOrientation, return_direction=False):
        super().__init__()
        self.nOrientation = nOrientation
        self.return_direction = return_direction

    def forward(self, input):
        return _RotationInvariantEncoding.apply(input, self.nOrientation, self.return_direction)


class _ActiveRotatingFilter(torch.autograd.Function):
    """models/orn/functions/active_rotating_filter.py:11-34"""

    @staticmethod
    def forward(ctx, input, indices):
        indices = indices.byte()
        ctx.save_for_backward(indices)
        return arf_forward(input, indices)

    @staticmethod
    def backward(ctx, grad_output):
        indices, = ctx.saved_tensors
        return arf_backward(indices, grad_output), None


active_rotating_filter = _ActiveRotatingFilter.apply


class ORConv2d(nn.Conv2d):
    """Conv2d whose weight is the ARF expansion of a [O,I,nOri,k,k] filter bank.

    At inference the expanded filter depends on the parameters only, so it is computed once
    and cached (the reference recomputes it on every forward of every FPN level,
    ORConv.py:80-82); the cache is invalidated whenever the weight tensor changes
    (version counter / dtype / device)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, arf_config=None, stride=1,
                 padding=0, dilation=1, groups=1, bias=True):
        self.nOrientation, self.nRotation = _pair(arf_config)
        for v, name in ((self.nOrientation, "nOrientation"), (self.nRotation, "nRotation")):
            assert v >= 1 and (v & (v - 1)) == 0, "invalid {} {}".format(name, v)
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.register_buffer("indices", self.get_indices())
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, self.nOrientation,
                                               *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels * self.nRotation))
        self.reset_parameters()
        self._arf_cache = None

    def reset_parameters(self):
        n = self.in_channels * getattr(self, "nOrientation", 1)
        for k in self.kernel_size:
            n *= k
        self.weight.data.normal_(0, math.sqrt(2.0 / n))
        if self.bias is not None:
            self.bias.data.zero_()

    def get_indices(self):
        kH, kW = self.kernel_size
        table = _ROT3 if kW == 3 else _ROT1
        d_ori, d_rot = 360 / self.nOrientation, 360 / self.nRotation
        idx = torch.zeros(self.nOrientation * kH * kW, self.nRotation, dtype=torch.uint8)
        for i in range(self.nOrientation):
            for j in range(kH * kW):
                for k in range(self.nRotation):
                    angle = d_rot * k
                    layer = (i + math.floor(angle / d_ori)) % self.nOrientation
                    idx[i * kH * kW + j, k] = int(layer * kH * kW + table[int(angle)][j])
        return idx.view(self.nOrientation, kH, kW, self.nRotation)

    def rotate_arf(self):
        w = self.weight
        key = (w._version, w.dtype, w.device, w.data_ptr())
        if self.training or torch.is_grad_enabled() and w.requires_grad:
            return active_rotating_filter(w, self.indices)
        if self._arf_cache is None or self._arf_cache[0] != key:
            e = arf_forward(w.detach(), self.indices)
            if getattr(self, "channels_last", False):
                e = e.contiguous(memory_format=torch.channels_last)
            self._arf_cache = (key, e)
        return self._arf_cache[1]

    def forward(self, input):
        w = self.rotate_arf()
        if input.is_cuda and not torch.is_grad_enabled():
            from .fused import own_conv_ok, conv_f16, PackedWeightCache
            if own_conv_ok(input, w.shape[1], w.shape[0], w.shape[2:], self.stride, self.padding, self.dilation,
                           self.groups):
                if not hasattr(self, "_packed"):
                    self._packed = PackedWeightCache()
                return conv_f16(input, self._packed.get(w), self.bias, w.shape[0], w.shape[2], 1, False)
        if input.is_cuda and self.bias is not None and not torch.is_grad_enabled():
            from .fused import bias_act_
            y = F.conv2d(input, w, None, self.stride, self.padding, self.dilation, self.groups)
            return bias_act_(y, self.bias, None, False)      # one-pass bias epilogue
        return F.conv2d(input, w, self.bias, self.stride, self.padding, self.dilation, self.groups)


def rot_inv_pool(x, n_orientation=8):
    _lib.require_cuda(x)
    N, c, h, w = x.shape
    if c % n_orientation:
        raise RuntimeError("channels must be a multiple of nOrientation")
    nhwc = x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
    if not nhwc:
        x = x.contiguous()
    out = torch.empty((N, c // n_orientation, h, w), dtype=x.dtype, device=x.device,
                      memory_format=torch.channels_last if nhwc else torch.contiguous_format)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().s2a_rot_inv_pool(
            _lib.ptr(x), N, c, h * w, n_orientation, _lib.dtype_code(x),
            _lib.LAYOUT_NHWC if nhwc else _lib.LAYOUT_NCHW, _lib.ptr(out), _lib.stream_ptr(x.device)))
    return out


class RotationInvariantPooling(nn.Module):
    def __init__(self, nInputPlane, nOrientation=8):
        super().__init__()
        self.nInputPlane = nInputPlane
        self.nOrientation = nOrientation

    def forward(self, x):
        return rot_inv_pool(x, self.nOrientation)
