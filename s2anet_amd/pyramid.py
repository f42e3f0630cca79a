"""Pyramid-packed head execution: every layer of S2ANetHead runs ONCE for all FPN levels.

The reference maps ``forward_single`` over the five levels (models/head.py:261-265): 14 layers x 5
launches, of which the P5-P7 ones (1024 / 256 / 64 positions) cannot fill 256 CUs.  The head's
filters are shared over the levels, so here the levels sit back to back in one channels-last
buffer ``[sum_l B*H_l*W_l, C]`` (level l = ``[B,H_l,W_l,C]`` at pixel offset ``pix0[l]``) and each
layer is one launch whose workgroups look up their level (s2a_*_pyramid_f16; 1x1 layers are plain
row GEMMs and need no table).  Same arithmetic as the per-level path, fewer and fuller launches.
"""
import ctypes

import torch

from . import _lib


class PyramidLayout:
    """level table + pixel offsets of a packed buffer for a given batch"""

    def __init__(self, batch, sizes, strides):
        assert 1 <= len(sizes) <= 8 and len(sizes) == len(strides)
        self.batch, self.sizes, self.strides = int(batch), [tuple(map(int, s)) for s in sizes], list(strides)
        self.c = _lib.Pyramid()
        self.c.n_levels = len(sizes)
        self.pix0, p = [], 0
        for i, ((h, w), s) in enumerate(zip(self.sizes, strides)):
            self.c.height[i], self.c.width[i], self.c.stride[i] = h, w, float(s)
            self.pix0.append(p)
            p += self.batch * h * w
        self.pixels = p

    def key(self):
        return (self.batch, tuple(self.sizes), tuple(self.strides))

    def new(self, channels, device, dtype=torch.float16):
        return torch.empty((self.pixels, channels), dtype=dtype, device=device)

    def level(self, buf, l, channels=None):
        """[B,C,H,W]-shaped (channels-last strided) view of level l of a packed buffer; channels = the
        leading columns to expose (narrow prediction heads live in 64-column buffers)"""
        h, w = self.sizes[l]
        v = buf[self.pix0[l]:self.pix0[l] + self.batch * h * w].view(self.batch, h, w, buf.shape[1])
        if channels is not None:
            v = v[..., :channels]
        return v.permute(0, 3, 1, 2)

    def rows(self, buf, l):
        h, w = self.sizes[l]
        return buf[self.pix0[l]:self.pix0[l] + self.batch * h * w]


def conv3x3(layout, x, packed_w, bias, out_channels, relu, residual=None, out=None):
    """3x3 / stride 1 / pad 1 on every level: x[P,C] -> out[P,O] (O multiple of 64)"""
    L = _lib.lib()
    if out is None:
        out = layout.new(out_channels, x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.s2a_conv3x3_pyramid_f16(_lib.ptr(x), _lib.ptr(packed_w), _lib.ptr(bias), _lib.ptr(residual),
                                             _lib.ptr(out), layout.batch, x.shape[1], out_channels, int(bool(relu)),
                                             ctypes.byref(layout.c), _lib.stream_ptr(x.device)))
    return out


def conv3x3_head(layout, x, packed_w, bias, out_channels, head_w, head_b, relu=True, keep_tower=False):
    """last 3x3 tower layer + its 1x1 prediction head in one launch -> head_out[P,64] (first <= 32 columns valid)
    (and the tower output [P,O] when keep_tower)"""
    L = _lib.lib()
    head_out = layout.new(64, x.device)
    tower = layout.new(out_channels, x.device) if keep_tower else None
    with torch.cuda.device(x.device):
        _lib.check(L.s2a_conv3x3_head_pyramid_f16(_lib.ptr(x), _lib.ptr(packed_w), _lib.ptr(bias), _lib.ptr(tower), _lib.ptr(head_w),
                                                  _lib.ptr(head_b), _lib.ptr(head_out), layout.batch, x.shape[1], out_channels,
                                                  int(bool(relu)), ctypes.byref(layout.c), _lib.stream_ptr(x.device)))
    return (head_out, tower) if keep_tower else head_out


def orconv_pool(layout, x, packed_w, bias, out_channels, n_orientation=8):
    """ORConv2d (cached ARF filter) + orientation max-pool in one launch: -> (out[P,O], pooled[P,O/8])"""
    assert n_orientation == 8
    L = _lib.lib()
    out = layout.new(out_channels, x.device)
    pooled = layout.new(out_channels // 8, x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.s2a_orconv_pool_pyramid_f16(_lib.ptr(x), _lib.ptr(packed_w), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(pooled),
                                                 layout.batch, x.shape[1], out_channels, ctypes.byref(layout.c),
                                                 _lib.stream_ptr(x.device)))
    return out, pooled


def conv3x3_wino(layout, x, packed_wino, bias, out_channels, relu, pool=False, out=None):
    """3x3 / stride 1 / pad 1 on every level in the Winograd F(2,3)-along-x form (s2a_conv3x3_wino_pyramid_f16):
    x[P,C] -> out[P,O]; pool: also the orientation max-pool [P,O/8] of the result (ORConv2d + RotationInvariantPooling)"""
    L = _lib.lib()
    if out is None:
        out = layout.new(out_channels, x.device)
    pooled = layout.new(out_channels // 8, x.device) if pool else None
    with torch.cuda.device(x.device):
        _lib.check(L.s2a_conv3x3_wino_pyramid_f16(_lib.ptr(x), _lib.ptr(packed_wino), _lib.ptr(bias), _lib.ptr(out),
                                                  _lib.ptr(pooled), layout.batch, x.shape[1], out_channels, int(bool(relu)),
                                                  ctypes.byref(layout.c), _lib.stream_ptr(x.device)))
    return (out, pooled) if pool else out


def wino_enabled():
    """S2A_CONV_WINO=1: the regular 256 -> 256 layers of the head on the Winograd F(2,3) kernel instead of the direct one.
    Off by default: measured level with the direct kernel on the same box (profiles/r06_wino_ab.txt), and the direct
    kernel's outputs are the ones the stage hashes of tests/test_net_forward.py pin."""
    import os
    return os.environ.get("S2A_CONV_WINO", "0") == "1"


def conv1x1(x, packed_w, bias, out_channels, relu, residual=None):
    """1x1 on packed rows (no geometry): x[P,C] -> out[P,O]"""
    L = _lib.lib()
    out = torch.empty((x.shape[0], out_channels), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.s2a_conv_nhwc_f16(_lib.ptr(x), _lib.ptr(packed_w), _lib.ptr(bias), _lib.ptr(residual), _lib.ptr(out),
                                       1, x.shape[1], 1, x.shape[0], out_channels, 1, 1, int(bool(relu)),
                                       _lib.stream_ptr(x.device)))
    return out


def align_conv(layout, x, anchors, packed_w, out_channels, relu=True):
    """fused AlignConv on every level: x[P,C] f16, anchors[P,5] f32, packed_w = alignconv.pack_weight"""
    L = _lib.lib()
    out = layout.new(out_channels, x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.s2a_align_conv_pyramid_f16(_lib.ptr(x), _lib.ptr(anchors), _lib.ptr(packed_w), _lib.ptr(out),
                                                layout.batch, x.shape[1], out_channels, int(bool(relu)),
                                                ctypes.byref(layout.c), _lib.stream_ptr(x.device)))
    return out


def fam_refine_anchors(layout, pred, anchor_scale):
    """grid anchors + FAM decode on every level: pred[P,>=5] f16 (first 5 columns) -> anchors[P,5] f32"""
    L = _lib.lib()
    out = torch.empty((layout.pixels, 5), dtype=torch.float32, device=pred.device)
    with torch.cuda.device(pred.device):
        _lib.check(L.s2a_fam_refine_anchors_pyramid(_lib.ptr(pred), pred.shape[1], layout.batch, ctypes.byref(layout.c),
                                                    float(anchor_scale), _lib.ptr(out), _lib.stream_ptr(pred.device)))
    return out


def rot_inv_pool(x, n_orientation=8):
    """orientation max on packed rows: x[P,C] -> [P,C/n]"""
    L = _lib.lib()
    out = torch.empty((x.shape[0], x.shape[1] // n_orientation), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.s2a_rot_inv_pool(_lib.ptr(x), 1, x.shape[1], x.shape[0], n_orientation, _lib.dtype_code(x),
                                      _lib.LAYOUT_NHWC, _lib.ptr(out), _lib.stream_ptr(x.device)))
    return out


def candidates(layout, cls, reg, anchors, num_classes, max_per_level=2000, wh_ratio_clip=16 / 1000):
    """get_bboxes' candidate selection for the whole batch (s2a_pyramid_candidates): packed predictions ->
    bboxes[B,n,5] f32, scores[B,n,C] f32; None when a level is too large for the fused top-k"""
    L = _lib.lib()
    n = L.s2a_pyramid_candidates_count(ctypes.byref(layout.c), int(max_per_level))
    if n < 0 or any(h * w > max_per_level > 0 and h * w > 24576 for h, w in layout.sizes):
        return None
    B = layout.batch
    bboxes = torch.empty((B, n, 5), dtype=torch.float32, device=cls.device)
    scores = torch.empty((B, n, num_classes), dtype=torch.float32, device=cls.device)
    sel = torch.empty((B, n), dtype=torch.int32, device=cls.device)
    with torch.cuda.device(cls.device):
        _lib.check(L.s2a_pyramid_candidates(_lib.ptr(cls), _lib.ptr(reg), _lib.ptr(anchors), B, ctypes.byref(layout.c),
                                            int(num_classes), int(max_per_level), float(wh_ratio_clip), _lib.ptr(bboxes),
                                            _lib.ptr(scores), _lib.ptr(sel), _lib.stream_ptr(cls.device)))
    return bboxes, scores, sel
