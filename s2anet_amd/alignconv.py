"""AlignConv (models/alignconv.py): anchor-driven deformable 3x3 conv + ReLU.

``AlignConv(in, out, kernel_size=3, deformable_groups=1).forward(x, anchors[B,H,W,5], stride)``
with sub-module name ``deform_conv`` (state-dict key ``deform_conv.weight``).

forward() runs the fully fused kernel (anchors -> sampling points -> MFMA contraction -> ReLU,
the [B,18,H,W] offset tensor is never materialised).  ``get_offset`` is kept with the reference
signature for callers that want the boundary tensor.
"""
import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

from . import _lib
from .dcn import DeformConv, _is_nhwc


def align_offsets(anchors, featmap_size, stride, kernel_size=3):
    """batched get_offset: anchors[B,H*W,5] or [B,H,W,5] (f32) -> [B,2*k*k,H,W] f32"""
    _lib.require_cuda(anchors)
    H, W = featmap_size
    a = anchors.reshape(-1, H * W, 5).float().contiguous()
    B = a.shape[0]
    out = torch.empty((B, 2 * kernel_size * kernel_size, H, W), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().s2a_align_offsets(_lib.ptr(a), B, H, W, float(stride), int(kernel_size),
                                                _lib.ptr(out), _lib.stream_ptr(a.device)))
    return out


def pack_weight(weight, dtype):
    """weight[O,C,3,3] -> the stage-major layout the fused kernel streams (s2a_dcn_pack_weight)"""
    _lib.require_cuda(weight)
    w = weight.detach().to(dtype).contiguous()
    L = _lib.lib()
    out = torch.empty((L.s2a_dcn_packed_elems(w.shape[0], w.shape[1], _lib.dtype_code(w)),), dtype=w.dtype, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.lib().s2a_dcn_pack_weight(_lib.ptr(w), w.shape[0], w.shape[1], _lib.dtype_code(w),
                                                  _lib.ptr(out), _lib.stream_ptr(w.device)))
    return out


def align_conv_forward(x, anchors, weight, stride, relu=True, packed=False, out_channels=None):
    """fused AlignConv: x[B,C,H,W] (NCHW or channels_last), anchors[B,H,W,5] f32, weight[O,C,3,3]
    (or, with packed=True, the output of pack_weight)"""
    _lib.require_cuda(x, anchors, weight)
    if x.dim() != 4:
        raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(x.dim()))
    nhwc = _is_nhwc(x)
    xx = x if nhwc else x.contiguous()
    B, C, H, W = xx.shape
    w = weight.contiguous()
    if w.dtype != xx.dtype:
        w = w.to(xx.dtype)
    O = out_channels if packed else w.shape[0]
    a = anchors.reshape(B, H, W, 5).float().contiguous()
    out = torch.empty((B, O, H, W), dtype=xx.dtype, device=xx.device,
                      memory_format=torch.channels_last if nhwc else torch.contiguous_format)
    p = _lib.AlignParams(B, C, H, W, O, float(stride), _lib.dtype_code(xx),
                         _lib.LAYOUT_NHWC if nhwc else _lib.LAYOUT_NCHW, int(bool(relu)), int(bool(packed)))
    L = _lib.lib()
    with torch.cuda.device(xx.device):
        ws = _lib.workspace(L.s2a_align_conv_workspace_bytes(p), xx.device, "dcn")
        _lib.check(L.s2a_align_conv_forward(_lib.ptr(xx), _lib.ptr(a), _lib.ptr(w), _lib.ptr(out), p,
                                            _lib.ptr(ws), ws.numel(), _lib.stream_ptr(xx.device)))
    return out


class AlignConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, deformable_groups=1):
        super().__init__()
        self.kernel_size = _pair(kernel_size)
        self.padding = tuple((s - 1) // 2 for s in self.kernel_size)
        self.deform_conv = DeformConv(in_channels, out_channels, kernel_size=self.kernel_size,
                                      padding=self.padding, deformable_groups=deformable_groups)
        self.relu = nn.ReLU(inplace=True)
        self._packed = None

    def packed_weight(self, dtype):
        """inference-time cache of the packed filter (invalidated when the parameter changes)"""
        w = self.deform_conv.weight
        key = (w._version, w.data_ptr(), dtype, w.device)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, pack_weight(w, dtype))
        return self._packed[1]

    def init_weights(self):
        nn.init.normal_(self.deform_conv.weight, 0, 0.01)   # alignconv.py:25-26

    @torch.no_grad()
    def get_offset(self, anchors, featmap_size, stride):
        """reference signature: anchors[H*W,5] of ONE image -> [18,H,W]"""
        return align_offsets(anchors[None], featmap_size, stride, self.kernel_size[0])[0]

    def fused_ok(self, x):
        kc = 32 if x.dtype == torch.float32 else 64
        return (self.kernel_size == (3, 3) and self.deform_conv.deformable_groups == 1 and
                self.deform_conv.groups == 1 and x.shape[1] % kc == 0 and
                self.deform_conv.out_channels % 64 == 0 and x.shape[2] >= 3 and x.shape[3] >= 3)

    def forward(self, x, anchors, stride):
        num_imgs, H, W = anchors.shape[:3]
        if self.fused_ok(x):
            if not (torch.is_grad_enabled() and self.deform_conv.weight.requires_grad):
                return align_conv_forward(x, anchors, self.packed_weight(x.dtype), stride, relu=True, packed=True,
                                          out_channels=self.deform_conv.out_channels)
            return align_conv_forward(x, anchors, self.deform_conv.weight, stride, relu=True)
        offset = align_offsets(anchors.reshape(num_imgs, H * W, 5), (H, W), stride, self.kernel_size[0])
        return self.relu(self.deform_conv(x, offset))
