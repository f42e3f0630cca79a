"""Rotated-box IoU and NMS — host side of the reference surface

    utils/box_iou_rotated/__init__.py          box_iou_rotated(boxes1, boxes2)
    utils/nms_rotated/__init__.py:6-11         nms_rotated(dets[N,6], iou_thr)
    utils/ml_nms_rotated/__init__.py           ml_nms_rotated(dets, scores, labels, iou_thr)
    utils/bbox_nms_rotated.py:5-64             multiclass_nms_rotated(...)

plus the pybind-shaped raw entry points used by ``s2anet_amd.compat`` and a batched
multiclass NMS (one launch sequence for a whole batch of images) used by the detector.
"""
import ctypes

import torch

from . import _lib


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


# ------------------------------------------------------------------ pybind-shaped entry points
def box_iou_rotated(boxes1, boxes2):
    """box_iou_rotated(boxes1[N,5], boxes2[M,5]) -> [N,M] float32
    (utils/box_iou_rotated/src/box_iou_rotated.h:22-42).  The reference reads raw
    float pointers (contiguous f32 required, models/utils.py:51-56); non-f32 /
    non-contiguous inputs are converted here instead of being mis-read."""
    _lib.require_cuda(boxes1, boxes2)
    if boxes1.device != boxes2.device:
        raise RuntimeError("boxes1 and boxes2 must be on the same device")
    b1, b2 = _f32c(boxes1), _f32c(boxes2)
    n, m = b1.shape[0], b2.shape[0]
    out = torch.empty((n, m), dtype=torch.float32, device=b1.device)
    if n == 0 or m == 0:
        return out
    if b1.shape[-1] != 5 or b2.shape[-1] != 5:
        raise RuntimeError("boxes must have 5 columns (x, y, w, h, angle)")
    L = _lib.lib()
    with torch.cuda.device(b1.device):
        nbytes = L.s2a_box_iou_rotated_workspace_bytes(n, m)
        ws = _lib.workspace(nbytes, b1.device, "iou")
        _lib.check(L.s2a_box_iou_rotated(_lib.ptr(b1), n, _lib.ptr(b2), m, _lib.ptr(out), _lib.ptr(ws),
                                         ws.numel(), _lib.stream_ptr(b1.device)))
    return out


def box_iou_rotated_pairs(boxes1, boxes2):
    _lib.require_cuda(boxes1, boxes2)
    b1, b2 = _f32c(boxes1), _f32c(boxes2)
    assert b1.shape == b2.shape and b1.shape[-1] == 5
    out = torch.empty((b1.shape[0],), dtype=torch.float32, device=b1.device)
    with torch.cuda.device(b1.device):
        _lib.check(_lib.lib().s2a_box_iou_rotated_pairs(_lib.ptr(b1), _lib.ptr(b2), b1.shape[0],
                                                        _lib.ptr(out), _lib.stream_ptr(b1.device)))
    return out


def polyiou_pairs(polys1, polys2):
    """polyiou.iou_poly element-wise: polys[n,8] (x1,y1..x4,y4) -> float64[n]
    (DOTA_devkit/polyiou/csrc/polyiou.cpp:108-128)"""
    _lib.require_cuda(polys1, polys2)
    p = polys1.to(torch.float64).contiguous().reshape(-1, 8)
    q = polys2.to(torch.float64).contiguous().reshape(-1, 8)
    assert p.shape == q.shape
    out = torch.empty((p.shape[0],), dtype=torch.float64, device=p.device)
    with torch.cuda.device(p.device):
        _lib.check(_lib.lib().s2a_polyiou_pairs(_lib.ptr(p), _lib.ptr(q), p.shape[0], _lib.ptr(out),
                                                _lib.stream_ptr(p.device)))
    return out


def assign_labels(anchors, gt_boxes, imgs_size=(1024, 1024), pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou_thr=0,
                  gt_max_assign_all=True, filter_invalid_anchors=True, filter_invalid_ious=True):
    """models/utils.py:33-147 in one call (IoU matrix in the workspace, three streaming passes over it):
    anchors[M,5], gt_boxes[N,5] (px / rad) -> assign_gt_ids[M] int64: -2 ignore, -1 negative, >= 0 gt index"""
    _lib.require_cuda(anchors, gt_boxes)
    a = anchors.float().contiguous()
    g = gt_boxes.float().contiguous().reshape(-1, 5)
    M, N = a.shape[0], g.shape[0]
    out = torch.full((M,), -2, dtype=torch.int64, device=a.device)
    if M == 0:
        return out
    L = _lib.lib()
    with torch.cuda.device(a.device):
        ws = _lib.workspace(L.s2a_assign_labels_workspace_bytes(M, N), a.device, "assign")
        _lib.check(L.s2a_assign_labels(_lib.ptr(a), M, _lib.ptr(g) if N else None, N, float(imgs_size[0]), float(imgs_size[1]),
                                       float(pos_iou_thr), float(neg_iou_thr), float(min_pos_iou_thr),
                                       int(bool(gt_max_assign_all)), int(bool(filter_invalid_anchors)),
                                       int(bool(filter_invalid_ious)), _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                       _lib.stream_ptr(a.device)))
    return out


def nms_poly(dets, thresh=0.5):
    """py_cpu_nms_poly_fast (DOTA_devkit/ResultMerge_multi_process.py:62-123) on the GPU:
    dets[n,9] = x1,y1,...,x4,y4,score -> kept indices (int64), descending score"""
    _lib.require_cuda(dets)
    d = dets.to(torch.float64).contiguous().reshape(-1, 9)
    n = d.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=d.device)
    L = _lib.lib()
    keep = torch.empty((n,), dtype=torch.int64, device=d.device)
    cnt = torch.empty((1,), dtype=torch.int64, device=d.device)
    host_k = ctypes.c_int64(0)
    with torch.cuda.device(d.device):
        ws = _lib.workspace(L.s2a_nms_poly_workspace_bytes(n), d.device, "nms_poly")
        _lib.check(L.s2a_nms_poly(_lib.ptr(d), n, float(thresh), _lib.ptr(keep), _lib.ptr(cnt), ctypes.byref(host_k),
                                  _lib.ptr(ws), ws.numel(), _lib.stream_ptr(d.device)))
    return keep[:host_k.value]


def _nms_raw_f64(dets, scores, labels, iou_threshold):
    """float64 boxes: the reference dispatches its NMS kernels on the dtype of `dets` (nms_rotated_cuda.cu:95-100) and
    evaluates single_box_iou_rotated<double> -- s2a_nms_rotated_f64 is that instantiation (keep decisions next to the
    threshold differ from the float32 evaluation; tests/golden/nms_f64.npz)"""
    d = dets.contiguous()
    s = scores.to(torch.float64).contiguous()
    lab = None if labels is None else labels.to(torch.float64).contiguous()
    n = d.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=d.device)
    if d.dim() != 2 or d.shape[1] != 5:
        raise RuntimeError("dets must be [N,5]")
    if s.numel() != n or (lab is not None and lab.numel() != n):
        raise RuntimeError("scores / labels must have N elements")
    L = _lib.lib()
    with torch.cuda.device(d.device):
        keep = torch.empty((n,), dtype=torch.int64, device=d.device)
        cnt = torch.empty((1,), dtype=torch.int64, device=d.device)
        ws = _lib.workspace(L.s2a_nms_rotated_f64_workspace_bytes(n), d.device, "nms_f64")
        host_k = ctypes.c_int64(0)
        _lib.check(L.s2a_nms_rotated_f64(_lib.ptr(d), _lib.ptr(s), _lib.ptr(lab), n, float(iou_threshold), _lib.ptr(keep),
                                         _lib.ptr(cnt), ctypes.byref(host_k), _lib.ptr(ws), ws.numel(),
                                         _lib.stream_ptr(d.device)))
    return keep[:host_k.value]


def _nms_raw(dets, scores, labels, iou_threshold):
    _lib.require_cuda(dets, scores, labels)
    if dets.dtype == torch.float64:
        return _nms_raw_f64(dets, scores, labels, iou_threshold)
    d = _f32c(dets)
    s = _f32c(scores)       # f16 scores are only sorted: f16->f32 is order preserving
    lab = None if labels is None else _f32c(labels)
    n = d.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=d.device)
    if d.dim() != 2 or d.shape[1] != 5:
        raise RuntimeError("dets must be [N,5]")
    if s.numel() != n or (lab is not None and lab.numel() != n):
        raise RuntimeError("scores / labels must have N elements")
    L = _lib.lib()
    with torch.cuda.device(d.device):
        keep = torch.empty((n,), dtype=torch.int64, device=d.device)
        cnt = torch.empty((1,), dtype=torch.int64, device=d.device)
        ws = _lib.workspace(L.s2a_nms_rotated_workspace_bytes(n, n), d.device, "nms")
        host_k = ctypes.c_int64(0)
        st = _lib.stream_ptr(d.device)
        if lab is None:
            rc = L.s2a_nms_rotated(_lib.ptr(d), _lib.ptr(s), n, float(iou_threshold), _lib.ptr(keep),
                                   _lib.ptr(cnt), ctypes.byref(host_k), _lib.ptr(ws), ws.numel(), st)
        else:
            rc = L.s2a_ml_nms_rotated(_lib.ptr(d), _lib.ptr(s), _lib.ptr(lab), n, float(iou_threshold),
                                      _lib.ptr(keep), _lib.ptr(cnt), ctypes.byref(host_k),
                                      _lib.ptr(ws), ws.numel(), st)
        _lib.check(rc)
    return keep[:host_k.value]


def nms_rotated_raw(dets, scores, iou_threshold):
    """nms_rotated_cuda.nms_rotated(dets[N,5], scores[N], thr) -> int64[K]
    (utils/nms_rotated/src/nms_rotated.h:21-41)"""
    return _nms_raw(dets, scores, None, iou_threshold)


def ml_nms_rotated(dets, scores, labels, iou_threshold):
    """ml_nms_rotated_cuda.ml_nms_rotated(dets[N,5], scores[N], labels[N], thr) -> int64[K]
    (utils/ml_nms_rotated/src/nms_rotated.h:23-44): keep indices, descending score."""
    return _nms_raw(dets, scores, labels, iou_threshold)


# ------------------------------------------------------------------ python wrappers of the reference
def nms_rotated(dets, iou_thr):
    """utils/nms_rotated/__init__.py:6-11 — dets[N,6] = x,y,w,h,a,score.  Returns ``dets`` alone
    when empty, ``(dets[keep], keep)`` otherwise (the reference's own quirk)."""
    if dets.shape[0] == 0:
        return dets
    keep_inds = nms_rotated_raw(dets[:, :5], dets[:, 5], iou_thr)
    return dets[keep_inds, :], keep_inds


def multiclass_nms_rotated(bboxes, scores, score_thr=0.05, iou_thr=0.5, max_per_img=2000):
    """utils/bbox_nms_rotated.py:5-64.  bboxes[n,5], scores[n,C] -> ([K,6], labels[K])."""
    num_classes = scores.size(1)
    assert bboxes.shape[1] == 5
    mask = scores > score_thr
    idx = mask.nonzero(as_tuple=False)             # row-major == boolean-mask order
    cand_scores = scores[mask]
    cand_boxes = bboxes[idx[:, 0]]
    labels = idx[:, 1].to(cand_boxes)              # float labels, as the reference (:40-42)
    if cand_boxes.shape[0] > 0:
        keep = ml_nms_rotated(cand_boxes, cand_scores, labels, iou_thr)
        cand_boxes, cand_scores, labels = cand_boxes[keep], cand_scores[keep], labels[keep]
        if keep.size(0) > max_per_img:
            _, inds = cand_scores.sort(descending=True)
            inds = inds[:max_per_img]
            cand_boxes, cand_scores, labels = cand_boxes[inds], cand_scores[inds], labels[inds]
        return torch.cat([cand_boxes, cand_scores[:, None].to(cand_boxes)], dim=1), labels
    out = bboxes.new_zeros((0, 6))
    return out, out.new_zeros((0, 1), dtype=torch.long)   # reference's inconsistent empty shape (:62-63)
    del num_classes


def batched_multiclass_nms_rotated(bboxes, scores, score_thr=0.05, iou_thr=0.5, max_per_img=2000,
                                   max_candidates=None, return_overflow=False, dropped_total=None, return_wire=False):
    """Whole-batch multiclass rotated NMS with NO host synchronisation and no stock tensor op.

    bboxes[B,n,5] f32, scores[B,n,C] -> dets[B,max_per_img,6] (x,y,w,h,a,score; zero padded),
    labels[B,max_per_img] int32 (-1 padded), counts[B] int32.  Same result per image as
    ``multiclass_nms_rotated`` (utils/bbox_nms_rotated.py:5-64) — candidates are
    (box, class) pairs with score > score_thr, suppressed per (image, class), survivors
    listed by descending score and truncated to max_per_img.

    Two library calls: ``s2a_multiclass_candidates`` (threshold + compaction) and
    ``s2a_nms_rotated_segmented_dets`` (NMS + the output rows, written once by its last kernel).  ``dets`` is a
    strided view of the wire buffer ``float32 [B, max_per_img*7 + 1]`` (rows x,y,w,h,a,score,label; last column the
    count) that the data-parallel detector all-gathers as it is (``return_wire=True`` appends it to the result).

    Static shapes: at most ``max_candidates`` (default n*C, i.e. lossless) candidates per
    batch are considered; the candidate list is compacted on the device.  The reference never
    drops a candidate (utils/bbox_nms_rotated.py:29-40), so a cap that is too small must not pass
    silently: ``return_overflow=True`` adds a result ``overflow`` = int64[2] on the device,
    ``[candidates found, candidates dropped]`` (dropped > 0 <=> the cap cut rows; still no host sync —
    the caller reads it when it synchronises anyway); ``dropped_total`` (a device int64 tensor of the caller)
    is incremented by the dropped count of this call, on the stream, by the same kernel.
    """
    _lib.require_cuda(bboxes, scores)
    B, n, C = scores.shape
    dev = bboxes.device
    bb = _f32c(bboxes).reshape(-1, 5)
    sc = _f32c(scores).reshape(-1)
    total = B * n * C
    if max_candidates is not None and int(max_candidates) < 1:
        raise ValueError("max_candidates must be >= 1 (a cap of zero would drop every candidate)")
    cap = total if max_candidates is None else min(int(max_candidates), total)
    L = _lib.lib()
    K = int(max_per_img)
    if total == 0:      # no box or no class: padded empty results through the same entry point (n = 0: no tensor is read)
        wire = torch.empty((B, K * 7 + 1), dtype=torch.float32, device=dev)
        labels = torch.empty((B, K), dtype=torch.int32, device=dev)
        counts = torch.empty((B,), dtype=torch.int32, device=dev)
        ncand = torch.zeros((1,), dtype=torch.int64, device=dev)
        overflow = torch.empty((2,), dtype=torch.int64, device=dev) if return_overflow else None
        if B > 0:
            with torch.cuda.device(dev):
                _lib.check(L.s2a_nms_rotated_segmented_dets(
                    None, None, None, None, None, 0, max(B * C, 1), B, float(iou_thr), K, _lib.ptr(wire), _lib.ptr(labels),
                    _lib.ptr(counts), _lib.ptr(ncand), _lib.ptr(overflow), _lib.ptr(dropped_total), None, 0,
                    _lib.stream_ptr(dev)))
        elif overflow is not None:
            overflow.zero_()
        out = (wire[:, :K * 7].view(B, K, 7)[..., :6], labels, counts)
        if return_overflow:
            out += (overflow,)
        return out + (wire,) if return_wire else out
    cboxes = torch.empty((cap, 5), dtype=torch.float32, device=dev)
    cscores = torch.empty((cap,), dtype=torch.float32, device=dev)
    seg = torch.empty((cap,), dtype=torch.int32, device=dev)
    grp = torch.empty((cap,), dtype=torch.int32, device=dev)
    cls = torch.empty((cap,), dtype=torch.int32, device=dev)
    ncand = torch.empty((1,), dtype=torch.int64, device=dev)
    wire = torch.empty((B, K * 7 + 1), dtype=torch.float32, device=dev)
    labels = torch.empty((B, K), dtype=torch.int32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    overflow = torch.empty((2,), dtype=torch.int64, device=dev) if return_overflow else None
    if dropped_total is not None:
        assert dropped_total.dtype == torch.int64 and dropped_total.device == dev and dropped_total.numel() == 1
    with torch.cuda.device(dev):
        st = _lib.stream_ptr(dev)
        ws = _lib.workspace(L.s2a_multiclass_candidates_workspace_bytes(total), dev, "cand")
        _lib.check(L.s2a_multiclass_candidates(_lib.ptr(bb), _lib.ptr(sc), B, n, C, float(score_thr), cap,
                                               _lib.ptr(cboxes), _lib.ptr(cscores), _lib.ptr(seg), _lib.ptr(grp),
                                               _lib.ptr(cls), _lib.ptr(ncand), _lib.ptr(ws), ws.numel(), st))
        # one (image, class) segment holds at most n rows: tight bound for the mask workspace
        ws2 = _lib.workspace(L.s2a_nms_rotated_workspace_bytes(cap, min(cap, n)), dev, "nms_b")
        _lib.check(L.s2a_nms_rotated_segmented_dets(
            _lib.ptr(cboxes), _lib.ptr(cscores), _lib.ptr(seg), _lib.ptr(grp), _lib.ptr(cls), cap, B * C, B,
            float(iou_thr), K, _lib.ptr(wire), _lib.ptr(labels), _lib.ptr(counts), _lib.ptr(ncand),
            _lib.ptr(overflow), _lib.ptr(dropped_total), _lib.ptr(ws2), ws2.numel(), st))
    dets = wire[:, :K * 7].view(B, K, 7)[..., :6]
    out = (dets, labels, counts)
    if return_overflow:
        out += (overflow,)
    if return_wire:
        out += (wire,)
    return out
