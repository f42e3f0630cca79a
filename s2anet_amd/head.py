"""S2ANet head, inference half (models/head.py:261-348, :648-725), batched.

Layer names and shapes follow the reference so that a reference ``state_dict`` loads unchanged
(``fam_reg_ls.N.0``, ``fam_cls_ls``, ``fam_reg_head``, ``fam_cls_head``, ``align_conv.deform_conv``,
``or_conv``, ``odm_reg_ls``, ``odm_cls_ls``, ``odm_cls_head``, ``odm_reg_head``).

What changed against the reference glue (same results, no per-image / per-level host work):
  * grid anchors are never built on the CPU and copied (head.py:315-326): the refined anchors
    come from ONE kernel per level (grid anchor + FAM decode fused, ``s2a_fam_refine_anchors``)
  * AlignConv consumes the refined anchors directly (no per-image get_offset loop, no offset tensor)
  * get_bboxes is batched: per-level top-k over the whole batch, one decode, ONE segmented
    rotated NMS launch sequence for all images, no host synchronisation
"""
import math
import os

import torch
import torch.nn as nn

from . import _lib
from .alignconv import AlignConv
from .orn import ORConv2d, RotationInvariantPooling
from .rotated import batched_multiclass_nms_rotated, multiclass_nms_rotated


def delta2bbox_rotated(rois, deltas, wh_ratio_clip=16 / 1000):
    """models/boxes.py:82-162 (is_encode_relative=True); rois/deltas [n,5] -> [n,5] f32"""
    _lib.require_cuda(rois, deltas)
    r, d = rois.float().contiguous(), deltas.float().contiguous()
    assert r.shape == d.shape and r.shape[-1] == 5
    out = torch.empty_like(r)
    n = r.numel() // 5
    with torch.cuda.device(r.device):
        _lib.check(_lib.lib().s2a_delta2bbox_rotated(_lib.ptr(r), _lib.ptr(d), n, float(wh_ratio_clip),
                                                     _lib.ptr(out), _lib.stream_ptr(r.device)))
    return out


rboxes_decode = delta2bbox_rotated   # models/boxes.py:223-247


def fam_refine_anchors(bbox_pred, stride, anchor_scale=4.0):
    """gen_grid_anchors (models/anchors.py:75-126) + fam_bbox_decode (models/head.py:27-52):
    bbox_pred[B,5,H,W] (f32/f16, NCHW or channels_last) -> refined anchors [B,H,W,5] f32"""
    _lib.require_cuda(bbox_pred)
    B, five, H, W = bbox_pred.shape
    assert five == 5
    nhwc = not bbox_pred.is_contiguous() and bbox_pred.is_contiguous(memory_format=torch.channels_last)
    p = bbox_pred if nhwc else bbox_pred.contiguous()
    out = torch.empty((B, H, W, 5), dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        _lib.check(_lib.lib().s2a_fam_refine_anchors(
            _lib.ptr(p), B, H, W, float(stride), float(anchor_scale), _lib.dtype_code(p),
            _lib.LAYOUT_NHWC if nhwc else _lib.LAYOUT_NCHW, _lib.ptr(out), _lib.stream_ptr(p.device)))
    return out


def _conv_relu(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=True),
                         nn.ReLU(inplace=True))


class PyramidPred(tuple):
    """the (fam_cls, fam_bbox, odm_cls, odm_bbox, refine_anchor) per-level lists of forward(), plus the
    pyramid-packed buffers they are views of (for the fused candidate selection)"""

    def __new__(cls, layout, odm_cls, odm_bbox, anchors, *lists):
        self = super().__new__(cls, lists)
        self.packed = (layout, odm_cls, odm_bbox, anchors)
        return self


class S2ANetHead(nn.Module):
    def __init__(self, num_classes, in_channels=256, feat_channels=256, stacked_convs=2,
                 with_orconv=True, anchor_scales=(4,), featmap_strides=(8, 16, 32, 64, 128),
                 score_thres_before_nms=0.05, iou_thres_nms=0.5, max_before_nms_per_level=2000,
                 max_per_img=2000, compute_fam_cls=True):
        super().__init__()
        assert len(anchor_scales) == 1, "S2ANet uses one square anchor per position (head.py:66-68)"
        self.num_classes = num_classes
        self.in_channels, self.feat_channels = in_channels, feat_channels
        self.stacked_convs, self.with_orconv = stacked_convs, with_orconv
        self.anchor_scale = float(anchor_scales[0])
        self.featmap_strides = tuple(featmap_strides)
        self.score_thres_before_nms = score_thres_before_nms
        self.iou_thres_nms = iou_thres_nms
        self.max_before_nms_per_level = max_before_nms_per_level
        self.max_per_img = max_per_img
        self.compute_fam_cls = compute_fam_cls      # the reference always evaluates it (head.py:306)
        fam_reg, fam_cls, odm_reg, odm_cls = [], [], [], []
        for i in range(stacked_convs):
            cin = in_channels if i == 0 else feat_channels
            fam_reg.append(_conv_relu(cin, feat_channels))
            fam_cls.append(_conv_relu(cin, feat_channels))
            odm_reg.append(_conv_relu(feat_channels, feat_channels))
            c0 = feat_channels // 8 if (i == 0 and with_orconv) else feat_channels
            odm_cls.append(_conv_relu(c0, feat_channels))
        self.fam_reg_ls, self.fam_cls_ls = nn.Sequential(*fam_reg), nn.Sequential(*fam_cls)
        self.fam_reg_head = nn.Conv2d(feat_channels, 5, kernel_size=(1, 1), padding=0, bias=True)
        self.fam_cls_head = nn.Conv2d(feat_channels, num_classes, kernel_size=(1, 1), padding=0, bias=True)
        self.align_conv = AlignConv(feat_channels, feat_channels, kernel_size=3)
        if with_orconv:
            self.or_conv = ORConv2d(feat_channels, feat_channels // 8, kernel_size=3, padding=1, arf_config=(1, 8))
            self.or_pool = RotationInvariantPooling(feat_channels, 8)
        else:
            self.or_conv = nn.Conv2d(feat_channels, feat_channels, 3, padding=1)
        self.odm_reg_ls, self.odm_cls_ls = nn.Sequential(*odm_reg), nn.Sequential(*odm_cls)
        self.odm_cls_head = nn.Conv2d(feat_channels, num_classes, kernel_size=(3, 3), padding=1, bias=True)
        self.odm_reg_head = nn.Conv2d(feat_channels, 5, kernel_size=(3, 3), padding=1, bias=True)
        self.init_weights()

    def init_weights(self):
        """head.py:230-258: N(0, 0.01) everywhere, classification biases = -log((1-p)/p), p = 0.01"""
        bias_cls = float(-math.log((1 - 0.01) / 0.01))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        self.align_conv.init_weights()
        nn.init.constant_(self.fam_cls_head.bias, bias_cls)
        nn.init.constant_(self.odm_cls_head.bias, bias_cls)

    # ------------------------------------------------------------------ forward
    def forward_single(self, x, stride):
        """one FPN level (head.py:296-348) -> (fam_cls, fam_bbox, odm_cls, odm_bbox, refine_anchor)"""
        fam_bbox_pred = self.fam_reg_head(self.fam_reg_ls(x))
        fam_cls_pred = self.fam_cls_head(self.fam_cls_ls(x)) if self.compute_fam_cls else None
        refine_anchor = fam_refine_anchors(fam_bbox_pred.detach(), stride, self.anchor_scale)   # [B,H,W,5]
        or_feat = self.or_conv(self.align_conv(x, refine_anchor, stride))
        odm_cls_feat = self.or_pool(or_feat) if self.with_orconv else or_feat
        odm_cls_pred = self.odm_cls_head(self.odm_cls_ls(odm_cls_feat))
        odm_bbox_pred = self.odm_reg_head(self.odm_reg_ls(or_feat))
        return fam_cls_pred, fam_bbox_pred, odm_cls_pred, odm_bbox_pred, refine_anchor

    # ------------------------------------------------------------------ all levels per launch
    def pyramid_ok(self, x):
        from .fused import FusedConv2d
        return (x.is_cuda and x.dtype == torch.float16 and not torch.is_grad_enabled() and self.with_orconv and
                isinstance(self.fam_reg_head, FusedConv2d) and self.in_channels % 64 == 0 and
                self.feat_channels % 64 == 0 and self.align_conv.kernel_size == (3, 3))

    def forward_pyramid(self, layout, x, anchors=None, trace=None):
        """forward_single for ALL FPN levels at once on a pyramid-packed feature buffer x[P,256]
        (s2anet_amd/pyramid.py).  Returns per-level lists of views with the forward_single shapes.
        anchors: refined anchors [P,5] f32 to sample with instead of the ones decoded from this call's own FAM
        regression (tests: separates the sampling from the regression that feeds it); trace: dict that receives
        the packed intermediate buffers by name (tests / scripts/f16_fixture_diag.py)"""
        from . import pyramid as P

        wino = P.wino_enabled()

        def tower(seq, t):
            for blk in seq:
                if wino and blk[0].wino_ok() and blk[0].in_channels >= 128:     # Winograd F(2,3) along x (wino_ops.hip)
                    w, b, o = blk[0].packed_args_wino()
                    t = P.conv3x3_wino(layout, t, w, b, o, relu=True)
                    continue
                w, b, o = blk[0].packed_args()
                t = P.conv3x3(layout, t, w, b, o, relu=True)
            return t

        def tower_with_head(seq, head, t):
            """the tower's last 3x3 layer and the 1x1 head that is its only reader: one launch"""
            fusable = (head.kernel_size == (1, 1) and head.out_channels <= 32 and seq[-1][0].out_channels == 256 and
                       seq[-1][0].in_channels % 64 == 0 and not os.environ.get("S2A_NO_FUSED_HEAD"))
            if not fusable:
                w, b, o = head.packed_args()
                return P.conv1x1(tower(seq, t), w, b, o, relu=False)
            t = tower(seq[:-1], t)
            w, b, o = seq[-1][0].packed_args()
            hw, hb, _ = head.packed_args()
            return P.conv3x3_head(layout, t, w, b, o, hw, hb, relu=True)

        fam_bbox = tower_with_head(self.fam_reg_ls, self.fam_reg_head, x)                   # [P,64], 5 used
        fam_cls = None
        if self.compute_fam_cls:
            fam_cls = tower_with_head(self.fam_cls_ls, self.fam_cls_head, x)
        own_anchors = P.fam_refine_anchors(layout, fam_bbox, self.anchor_scale)             # [P,5] f32
        if anchors is None:
            anchors = own_anchors
        else:
            assert anchors.shape == own_anchors.shape and anchors.dtype == torch.float32 and anchors.is_contiguous()
        if getattr(self, "capture", None) is not None:      # bench.py: the operands of this step's launches
            self.capture.update(layout=layout, x=x, anchors=anchors)
        al = P.align_conv(layout, x, anchors, self.align_conv.packed_weight(torch.float16), self.feat_channels)
        wa = self.or_conv.rotate_arf()
        if not hasattr(self.or_conv, "_packed"):
            from .fused import PackedWeightCache
            self.or_conv._packed = PackedWeightCache()
        if self.or_pool.nOrientation == 8 and wa.shape[0] % 64 == 0 and wino and wa.shape[1] % 32 == 0:
            or_feat, pooled = P.conv3x3_wino(layout, al, self.or_conv._packed.get_wino(wa),
                                             self.or_conv._packed.get_bias(self.or_conv.bias, wa.shape[0]), wa.shape[0],
                                             relu=False, pool=True)
        elif self.or_pool.nOrientation == 8 and wa.shape[0] % 64 == 0:                      # conv + orientation max, one launch
            or_feat, pooled = P.orconv_pool(layout, al, self.or_conv._packed.get(wa),
                                            self.or_conv._packed.get_bias(self.or_conv.bias, wa.shape[0]), wa.shape[0])
        else:
            or_feat = P.conv3x3(layout, al, self.or_conv._packed.get(wa),
                                self.or_conv._packed.get_bias(self.or_conv.bias, wa.shape[0]), wa.shape[0], relu=False)
            pooled = P.rot_inv_pool(or_feat, self.or_pool.nOrientation)                     # [P,32]
        w, b, o = self.odm_cls_head.packed_args()
        odm_cls = P.conv3x3(layout, tower(self.odm_cls_ls, pooled), w, b, o, relu=False)    # [P,64], C used
        w, b, o = self.odm_reg_head.packed_args()
        odm_bbox = P.conv3x3(layout, tower(self.odm_reg_ls, or_feat), w, b, o, relu=False)  # [P,64], 5 used
        n = len(layout.sizes)
        if trace is not None:
            trace.update(x=x, fam_bbox=fam_bbox, fam_cls=fam_cls, own_anchors=own_anchors, anchors=anchors, align=al,
                         or_feat=or_feat, pooled=pooled, odm_cls=odm_cls, odm_bbox=odm_bbox)
        return PyramidPred(layout, odm_cls, odm_bbox, anchors,
                           [layout.level(fam_cls, l, self.num_classes) for l in range(n)] if fam_cls is not None else [None] * n,
                [layout.level(fam_bbox, l, 5) for l in range(n)],
                [layout.level(odm_cls, l, self.num_classes) for l in range(n)],
                [layout.level(odm_bbox, l, 5) for l in range(n)],
                [layout.rows(anchors, l).view(layout.batch, *layout.sizes[l], 5) for l in range(n)])

    def forward(self, feats, post_process=False):
        per_level = [self.forward_single(f, s) for f, s in zip(feats, self.featmap_strides)]
        p = tuple(map(list, zip(*per_level)))
        results = {"loss": None, "loss_items": None, "boxes_ls": None, "pred": None}
        if post_process:
            results["boxes_ls"] = self.get_bboxes(p)
        else:
            results["pred"] = p
        return results

    # ------------------------------------------------------------------ decode + NMS, batched
    def candidates(self, p, raw_logits=False):
        """per-level sigmoid + top-k (head.py:697-705), levels concatenated (head.py:712-714),
        final decode (head.py:717).  -> bboxes[B,n,5] f32, scores[B,n,C] f32 with n <= 5344"""
        odm_cls, odm_bbox, anchors = p[2], p[3], p[4]
        k = self.max_before_nms_per_level
        sc_l, bb_l, an_l = [], [], []
        for cls, reg, anc in zip(odm_cls, odm_bbox, anchors):
            B = cls.shape[0]
            s = cls.detach().permute(0, 2, 3, 1).reshape(B, -1, self.num_classes)
            s = s.float() if raw_logits else s.sigmoid()   # raw_logits: calibration only (same ranking)
            d = reg.detach().permute(0, 2, 3, 1).reshape(B, -1, 5)
            a = anc.reshape(B, -1, 5)
            if k > 0 and s.shape[1] > k:
                top = s.max(dim=2)[0].topk(k, dim=1)[1]
                s = s.gather(1, top[..., None].expand(-1, -1, self.num_classes))
                d = d.gather(1, top[..., None].expand(-1, -1, 5))
                a = a.gather(1, top[..., None].expand(-1, -1, 5))
            sc_l.append(s.float())
            bb_l.append(d.float())
            an_l.append(a)
        scores, deltas, anc = torch.cat(sc_l, 1), torch.cat(bb_l, 1), torch.cat(an_l, 1)
        B, n = scores.shape[:2]
        bboxes = delta2bbox_rotated(anc.reshape(-1, 5), deltas.reshape(-1, 5)).reshape(B, n, 5)
        return bboxes, scores

    def get_bboxes_batched(self, p, max_candidates=None, return_overflow=False, **nms_kw):
        """-> dets[B,max_per_img,6], labels[B,max_per_img] (-1 padded), counts[B]; no host sync.
        return_overflow: fourth result int64[2] = [NMS candidates found, candidates dropped by max_candidates];
        nms_kw: dropped_total / return_wire of batched_multiclass_nms_rotated"""
        fused = None
        if isinstance(p, PyramidPred) and not os.environ.get("S2A_NO_FUSED_CANDIDATES"):
            from . import pyramid as P
            layout, cls, reg, anc = p.packed
            fused = P.candidates(layout, cls, reg, anc, self.num_classes, self.max_before_nms_per_level)
        bboxes, scores = fused[:2] if fused is not None else self.candidates(p)
        return batched_multiclass_nms_rotated(bboxes, scores, self.score_thres_before_nms,
                                              self.iou_thres_nms, self.max_per_img, max_candidates, return_overflow,
                                              **nms_kw)

    def get_bboxes(self, p):
        """reference return shape (head.py:648-682): list of (det_bboxes[K,6], det_labels[K]) per image"""
        bboxes, scores = self.candidates(p)
        return [multiclass_nms_rotated(bboxes[b], scores[b], self.score_thres_before_nms,
                                       self.iou_thres_nms, self.max_per_img)
                for b in range(bboxes.shape[0])]
