"""DOTA Task-1 evaluation with the polygon overlaps on the GPU.

``voc_eval_arrays`` is ``voc_eval`` of DOTA_devkit/dota_evaluation_task1.py:92-318 for ONE class on arrays
instead of files: the per-detection search for the best-overlapping ground truth (HBB prefilter + polyiou,
:204-263) is one kernel over all detections (s2a_polyiou_match); the greedy TP/FP marking in confidence order
(:265-290) and the AP integral (:58-89) are the reference's scalar bookkeeping on the host.
"""
import numpy as np
import torch

from . import _lib


def polyiou_match(det_polys, det_image, gt_polys, gt_offsets):
    """det_polys[D,8] f64, det_image[D] int32, gt_polys[G,8] f64 grouped by image, gt_offsets[I+1] int64
    -> ovmax[D] f64 (-inf: no overlapping gt), argmax[D] int64 (index into gt_polys or -1)"""
    _lib.require_cuda(det_polys, det_image, gt_polys, gt_offsets)
    d = det_polys.to(torch.float64).contiguous().reshape(-1, 8)
    g = gt_polys.to(torch.float64).contiguous().reshape(-1, 8)
    img = det_image.to(torch.int32).contiguous()
    off = gt_offsets.to(torch.int64).contiguous()
    D = d.shape[0]
    ov = torch.empty((D,), dtype=torch.float64, device=d.device)
    am = torch.empty((D,), dtype=torch.int64, device=d.device)
    with torch.cuda.device(d.device):
        _lib.check(_lib.lib().s2a_polyiou_match(_lib.ptr(d), _lib.ptr(img), D, _lib.ptr(g), _lib.ptr(off), off.numel() - 1,
                                                _lib.ptr(ov), _lib.ptr(am), _lib.stream_ptr(d.device)))
    return ov, am


def voc_ap(rec, prec, use_07_metric=False):
    """dota_evaluation_task1.py:58-89"""
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def mark_tp_fp(ovmax, argmax, gt_difficult, ovthresh=0.5, is_filter_difficult=True):
    """:265-290 on detections already in descending-confidence order"""
    n = len(ovmax)
    tp, fp = np.zeros(n), np.zeros(n)
    taken = np.zeros(len(gt_difficult), bool)
    for k in range(n):
        if ovmax[k] > ovthresh:
            j = argmax[k]
            if is_filter_difficult and gt_difficult[j]:
                continue
            if not taken[j]:
                tp[k] = 1.0
                taken[j] = True
            else:
                fp[k] = 1.0
        else:
            fp[k] = 1.0
    return tp, fp


def voc_eval_arrays(det_polys, det_scores, det_image, gt_polys, gt_image, gt_difficult, num_images, ovthresh=0.5,
                    is_filter_difficult=True, use_07_metric=False, device="cuda"):
    """one class: detections (polygons[D,8], confidences[D], image index[D]) against ground truth
    (polygons[G,8], image index[G], difficult[G]) -> rec, prec, ap, sorted_scores as voc_eval returns them"""
    det_polys = np.asarray(det_polys, np.float64).reshape(-1, 8)
    det_scores = np.asarray(det_scores, np.float64)
    det_image = np.asarray(det_image, np.int64)
    gt_polys = np.asarray(gt_polys, np.float64).reshape(-1, 8)
    gt_image = np.asarray(gt_image, np.int64)
    gt_difficult = np.asarray(gt_difficult).astype(bool)
    num_gts = int((~gt_difficult).sum()) if is_filter_difficult else int(gt_difficult.shape[0])
    if det_polys.shape[0] == 0:
        return np.zeros(1), np.zeros(1), 0.0, np.zeros(1)
    go = np.argsort(gt_image, kind="stable")                      # group the ground truth by image (file order kept)
    gt_polys, gt_image, gt_difficult = gt_polys[go], gt_image[go], gt_difficult[go]
    offsets = np.zeros(num_images + 1, np.int64)
    np.add.at(offsets, gt_image + 1, 1)
    offsets = np.cumsum(offsets)
    order = np.argsort(-det_scores)                                # :183
    dev = torch.device(device)
    ov, am = polyiou_match(torch.from_numpy(det_polys[order]).to(dev), torch.from_numpy(det_image[order].astype(np.int32)).to(dev),
                           torch.from_numpy(gt_polys).to(dev), torch.from_numpy(offsets).to(dev))
    tp, fp = mark_tp_fp(ov.cpu().numpy(), am.cpu().numpy(), gt_difficult, ovthresh, is_filter_difficult)
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(num_gts)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric), det_scores[order]
