"""CPU oracle for the S2ANet dense-inference hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / CPU baseline.  The product path
(``s2anet_amd``) never imports it.

Two layers:
  * ``libs2a_oracle.so``  (oracle/s2a_oracle.cpp): C++ restatement of the native ops
    (rotated IoU both sort branches, NMS / ml-NMS both rules, polyiou, ARF, rotation
    invariant pooling, deformable conv forward).
  * numpy float32 restatements of the reference's Python head glue (this file):
    grid anchors, rotated delta decode, AlignConv.get_offset, multiclass NMS.

Pinning status: see the header of s2a_oracle.cpp and tests/test_oracle_pinned.py.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libs2a_oracle.so")

SORT_CPU = 0   # std::sort branch (reference host build)
SORT_GPU = 1   # swap-sort branch (reference __CUDACC__ build)
RULE_GE = 0    # reference CPU NMS: suppress when iou >= thr
RULE_GT = 1    # reference GPU NMS: suppress when iou >  thr


def _src_sha():
    import hashlib
    with open(os.path.join(_HERE, "s2a_oracle.cpp"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def _gpu_or_profiler_live():
    """a process that has initialised the GPU (or that a GPU profiler is preloaded into) must not start child processes
    on this pool: then a missing / stale library is an error to report, not something to rebuild"""
    if os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("HSA_TOOLS_LIB") or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return True
    try:
        import sys
        torch = sys.modules.get("torch")
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        return False


def build(force=False):
    """compile libs2a_oracle.so when it is missing or its source changed.  Staleness is decided by the CONTENT of the
    source (sha256 kept beside the library), not by mtimes: a snapshot copied to the GPU box does not keep them."""
    stamp = _LIB_PATH + ".src_sha256"
    want = _src_sha()
    have = open(stamp).read().strip() if os.path.exists(stamp) else None
    if not force and os.path.exists(_LIB_PATH) and have == want:
        return _LIB_PATH
    if _gpu_or_profiler_live():
        if os.path.exists(_LIB_PATH) and have is None and not force:
            return _LIB_PATH            # a library built before the stamp existed: use it, never shell out from here
        raise RuntimeError("oracle/libs2a_oracle.so is missing or stale and this process has the GPU initialised: build it "
                           "first (python -c 'import oracle; oracle.build()' or __graft_entry__.build())")
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        f32p = ctypes.POINTER(ctypes.c_float)
        f64p = ctypes.POINTER(ctypes.c_double)
        i64p = ctypes.POINTER(ctypes.c_int64)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        i64, ci = ctypes.c_int64, ctypes.c_int
        L.orc_iou_single.restype = ctypes.c_float
        L.orc_iou_single.argtypes = [f32p, f32p, ci]
        L.orc_box_iou_rotated.restype = None
        L.orc_box_iou_rotated.argtypes = [f32p, i64, f32p, i64, f32p, ci, ci]
        L.orc_iou_pairs.restype = None
        L.orc_iou_pairs.argtypes = [f32p, f32p, i64, ci, f32p, ci]
        L.orc_iou_pairs_f64.restype = None
        L.orc_iou_pairs_f64.argtypes = [f64p, f64p, i64, ci, f64p, ci]
        L.orc_nms_rotated_f64.restype = i64
        L.orc_nms_rotated_f64.argtypes = [f64p, f64p, f64p, i64, ctypes.c_float, ci, ci, i64p]
        L.orc_nms_rotated.restype = i64
        L.orc_nms_rotated.argtypes = [f32p, f32p, f32p, i64, ctypes.c_float, ci, ci, ci, i64p]
        L.orc_nms_margin.restype = ctypes.c_double
        L.orc_nms_margin.argtypes = [f32p, f32p, i64, ctypes.c_float, ci]
        L.orc_polyiou.restype = ctypes.c_double
        L.orc_polyiou.argtypes = [f64p, f64p]
        L.orc_polyiou_pairs.restype = None
        L.orc_polyiou_pairs.argtypes = [f64p, f64p, i64, f64p]
        L.orc_nms_poly.restype = i64
        L.orc_nms_poly.argtypes = [f64p, i64, ctypes.c_double, i64p]
        L.orc_arf_forward.restype = None
        L.orc_arf_forward.argtypes = [f32p, u8p, i64, i64, ci, ci, ci, ci, f32p]
        L.orc_arf_backward.restype = None
        L.orc_arf_backward.argtypes = [f32p, u8p, i64, i64, ci, ci, ci, ci, f32p]
        L.orc_rot_inv_pool.restype = None
        L.orc_rot_inv_pool.argtypes = [f32p, i64, i64, i64, ci, f32p]
        L.orc_deform_conv_backward.restype = None
        L.orc_deform_conv_backward.argtypes = [f32p, f32p, f32p, f32p, i64, i64, i64, i64, i64] + [ctypes.c_int] * 9 + \
            [f32p, f32p, f32p]
        L.orc_deform_conv_forward.restype = None
        L.orc_deform_conv_forward.argtypes = [f32p, f32p, f32p, i64, i64, i64, i64, i64,
                                              ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, f32p]
        L.orc_deform_conv_forward_half.restype = None
        L.orc_deform_conv_forward_half.argtypes = [f32p, f32p, f32p, i64, i64, i64, i64, i64,
                                                   ci, ci, ci, ci, ci, ci, ci, ci, ci, f32p]
        L.orc_round_f16.restype = ctypes.c_float
        L.orc_round_f16.argtypes = [ctypes.c_float]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(t))


# --------------------------------------------------------------------------- native ops
def box_iou_rotated(boxes1, boxes2, sort_mode=SORT_GPU, cull=False):
    b1, b2 = _f32(boxes1).reshape(-1, 5), _f32(boxes2).reshape(-1, 5)
    out = np.empty((b1.shape[0], b2.shape[0]), np.float32)
    if out.size:
        lib().orc_box_iou_rotated(_p(b1), b1.shape[0], _p(b2), b2.shape[0], _p(out),
                                  sort_mode, int(cull))
    return out


def iou_pairs(b1, b2, sort_mode=SORT_GPU):
    b1, b2 = _f32(b1), _f32(b2)
    assert b1.shape == b2.shape and b1.shape[1] in (5, 6)
    out = np.empty(b1.shape[0], np.float32)
    lib().orc_iou_pairs(_p(b1), _p(b2), b1.shape[0], b1.shape[1], _p(out), sort_mode)
    return out


def nms_rotated(dets, scores, iou_thr, labels=None, rule=RULE_GT, sort_mode=SORT_GPU, cull=False):
    d, s = _f32(dets).reshape(-1, 5), _f32(scores).reshape(-1)
    n = d.shape[0]
    keep = np.empty(n, np.int64)
    lab = None if labels is None else _f32(labels).reshape(-1)
    k = lib().orc_nms_rotated(_p(d), _p(s), None if lab is None else _p(lab), n,
                              float(iou_thr), rule, sort_mode, int(cull),
                              _p(keep, ctypes.c_int64))
    return keep[:k].copy()


def iou_pairs_f64(b1, b2, sort_mode=SORT_GPU):
    """single_box_iou_rotated<double> element-wise: b1/b2 [n,5] (or [n,6] with the label test) float64 -> float64[n]"""
    b1 = np.ascontiguousarray(b1, np.float64)
    b2 = np.ascontiguousarray(b2, np.float64)
    out = np.empty(b1.shape[0], np.float64)
    lib().orc_iou_pairs_f64(_p(b1, ctypes.c_double), _p(b2, ctypes.c_double), b1.shape[0], b1.shape[1],
                            _p(out, ctypes.c_double), sort_mode)
    return out


def nms_rotated_f64(dets, scores, iou_thr, labels=None, rule=RULE_GT, sort_mode=SORT_GPU):
    """nms_rotated / ml_nms_rotated on float64 boxes (the ops dispatch on dets' dtype): keep indices, descending score"""
    d = np.ascontiguousarray(dets, np.float64)
    s = np.ascontiguousarray(scores, np.float64)
    lab = None if labels is None else np.ascontiguousarray(labels, np.float64)
    keep = np.empty(d.shape[0], np.int64)
    k = lib().orc_nms_rotated_f64(_p(d, ctypes.c_double), _p(s, ctypes.c_double),
                                  None if lab is None else _p(lab, ctypes.c_double), d.shape[0], float(iou_thr), rule,
                                  sort_mode, keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return keep[:k].copy()


def ml_nms_rotated(dets, scores, labels, iou_thr, **kw):
    return nms_rotated(dets, scores, iou_thr, labels=labels, **kw)


def nms_margin(dets, labels, iou_thr, sort_mode=SORT_GPU):
    d = _f32(dets).reshape(-1, 5)
    lab = None if labels is None else _f32(labels).reshape(-1)
    return lib().orc_nms_margin(_p(d), None if lab is None else _p(lab), d.shape[0],
                                float(iou_thr), sort_mode)


def polyiou(p8, q8):
    p = np.ascontiguousarray(p8, np.float64).reshape(-1, 8)
    q = np.ascontiguousarray(q8, np.float64).reshape(-1, 8)
    out = np.empty(p.shape[0], np.float64)
    lib().orc_polyiou_pairs(_p(p, ctypes.c_double), _p(q, ctypes.c_double), p.shape[0],
                            _p(out, ctypes.c_double))
    return out


def rbox_to_poly(boxes):
    """rotated_box_to_poly_single (utils/general.py:886-921) with cv2.boxPoints restated from OpenCV 4.x
    RotatedRect::points (OpenCV is absent here: parity unpinned for this helper) -> [N,8] float32"""
    out = np.empty((len(boxes), 8), np.float32)
    for i, r in enumerate(np.asarray(boxes, np.float32)):
        x, y, w, h = (np.float32(v) for v in r[:4])
        angle = float(r[4])
        if angle < 0:
            angle += math.pi
        angle = (angle / math.pi) * 180
        e1, e2 = (w, h)
        if angle > 90:
            angle, e1, e2 = angle - 90, h, w
        rad = float(np.float32(angle)) * math.pi / 180.0
        b, a = np.float32(math.cos(rad)) * np.float32(0.5), np.float32(math.sin(rad)) * np.float32(0.5)
        p0x, p0y = x - a * e2 - b * e1, y + b * e2 - a * e1
        p1x, p1y = x + a * e2 - b * e1, y - b * e2 - a * e1
        out[i] = [p0x, p0y, p1x, p1y, 2 * x - p0x, 2 * y - p0y, 2 * x - p1x, 2 * y - p1y]
    return out


def voc_eval_arrays(det_polys, det_scores, det_image, gt_polys, gt_image, gt_difficult, num_images, ovthresh=0.5,
                    is_filter_difficult=True, use_07_metric=False):
    """voc_eval (DOTA_devkit/dota_evaluation_task1.py:92-318) for one class on arrays, restated with the oracle's
    polyiou: -> rec, prec, ap, sorted_scores, (ovmax, argmax) per detection in confidence order"""
    det_polys = np.asarray(det_polys, np.float64).reshape(-1, 8)
    det_scores = np.asarray(det_scores, np.float64)
    det_image = np.asarray(det_image, np.int64)
    gt_polys = np.asarray(gt_polys, np.float64).reshape(-1, 8)
    gt_image = np.asarray(gt_image, np.int64)
    gt_difficult = np.asarray(gt_difficult).astype(bool)
    num_gts = int((~gt_difficult).sum()) if is_filter_difficult else int(gt_difficult.shape[0])
    order = np.argsort(-det_scores)
    taken = np.zeros(gt_polys.shape[0], bool)
    n = det_polys.shape[0]
    tp, fp, ovs, args = np.zeros(n), np.zeros(n), np.full(n, -np.inf), np.full(n, -1, np.int64)
    for k, d in enumerate(order):
        bb = det_polys[d]
        idx = np.where(gt_image == det_image[d])[0]
        ovmax, jmax = -np.inf, -1
        if idx.size:
            g = gt_polys[idx]
            gx1, gy1, gx2, gy2 = g[:, 0::2].min(1), g[:, 1::2].min(1), g[:, 0::2].max(1), g[:, 1::2].max(1)
            px1, py1, px2, py2 = bb[0::2].min(), bb[1::2].min(), bb[0::2].max(), bb[1::2].max()
            iw = np.maximum(np.minimum(gx2, px2) - np.maximum(gx1, px1) + 1.0, 0.0)
            ih = np.maximum(np.minimum(gy2, py2) - np.maximum(gy1, py1) + 1.0, 0.0)
            inters = iw * ih
            uni = (px2 - px1 + 1.0) * (py2 - py1 + 1.0) + (gx2 - gx1 + 1.0) * (gy2 - gy1 + 1.0) - inters
            keep = np.where(inters / uni > 0)[0]
            if keep.size:
                ov = polyiou(g[keep], np.repeat(bb[None], keep.size, 0))
                ovmax, jmax = ov.max(), idx[keep[int(np.argmax(ov))]]
        ovs[k], args[k] = ovmax, jmax
        if ovmax > ovthresh:
            if is_filter_difficult and gt_difficult[jmax]:
                continue
            if not taken[jmax]:
                tp[k], taken[jmax] = 1.0, True
            else:
                fp[k] = 1.0
        else:
            fp[k] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(num_gts)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            ap += (0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])) / 11.0
    else:
        mrec, mpre = np.concatenate(([0.0], rec, [1.0])), np.concatenate(([0.0], prec, [0.0]))
        for i in range(mpre.size - 1, 0, -1):
            mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
        i = np.where(mrec[1:] != mrec[:-1])[0]
        ap = np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])
    return rec, prec, ap, det_scores[order], (ovs, args)


def assign_labels(anchors, gt_boxes, imgs_size=(1024, 1024), pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou_thr=0,
                  gt_max_assign_all=True, filter_invalid_anchors=True, filter_invalid_ious=True, sort_mode=None):
    """models/utils.py:33-147 restated on top of the oracle's box_iou_rotated (sort_mode: SORT_CPU to compare
    with the reference's CPU op, SORT_GPU (default) for what its CUDA op computes)"""
    a = np.ascontiguousarray(anchors, np.float32).reshape(-1, 5)
    g = np.ascontiguousarray(gt_boxes, np.float32).reshape(-1, 5)
    M, N = a.shape[0], g.shape[0]
    out = np.full(M, -2, np.int64)
    flags = np.ones(M, bool)
    if filter_invalid_anchors:                                                      # :63-69
        flags = (a[:, 0] >= 0) & (a[:, 1] >= 0) & (a[:, 0] <= imgs_size[1]) & (a[:, 1] <= imgs_size[0]) & \
                (a[:, 2] < imgs_size[1]) & (a[:, 3] < imgs_size[0])
    if N == 0:                                                                      # :72-80
        out[flags] = -1
        return out
    ious = box_iou_rotated(a, g, sort_mode=SORT_GPU if sort_mode is None else sort_mode).copy()
    if filter_invalid_ious:                                                         # :86-93
        ious[~((ious >= 0) & (ious <= 1))] = -0.5
    ious[~flags] = -0.5                                                             # :97-98
    max_ious, argmax = ious.max(1), ious.argmax(1)                                  # :107 (first index)
    out[(max_ious >= 0) & (max_ious < neg_iou_thr)] = -1
    pos = max_ious >= pos_iou_thr
    out[pos] = argmax[pos]
    gt_max, gt_arg = ious.max(0), ious.argmax(0)                                    # :121
    for i in range(N):                                                              # :123-143
        if gt_max[i] > min_pos_iou_thr:
            if gt_max_assign_all:
                out[ious[:, i] == gt_max[i]] = i
            else:
                out[gt_arg[i]] = i
    return out


def nms_poly(dets9, thresh=0.5):
    """py_cpu_nms_poly_fast restated: dets[n,9] float64 -> kept original indices, score descending"""
    d = np.ascontiguousarray(dets9, np.float64).reshape(-1, 9)
    keep = np.empty(d.shape[0], np.int64)
    k = lib().orc_nms_poly(_p(d, ctypes.c_double), d.shape[0], float(thresh), _p(keep, ctypes.c_int64))
    return keep[:k].copy()


def arf_forward(weight, indices):
    w = _f32(weight)
    idx = np.ascontiguousarray(indices, np.uint8)
    O, I, nOri, kH, kW = w.shape
    nRot = idx.shape[3]
    out = np.empty((O * nRot, I * nOri, kH, kW), np.float32)
    lib().orc_arf_forward(_p(w), _p(idx, ctypes.c_uint8), O, I, nOri, kH, kW, nRot, _p(out))
    return out


def arf_backward(indices, grad_output):
    """grad_output [O*nRot, I*nOri, kH, kW] -> grad_input [O, I, nOri, kH, kW]"""
    idx = np.ascontiguousarray(indices, np.uint8)
    g = _f32(grad_output)
    nOri, kH, kW, nRot = idx.shape
    O, I = g.shape[0] // nRot, g.shape[1] // nOri
    out = np.empty((O, I, nOri, kH, kW), np.float32)
    lib().orc_arf_backward(_p(g), _p(idx, ctypes.c_uint8), O, I, nOri, kH, kW, nRot, _p(out))
    return out


def rie_forward(feature, n_ori):
    """RIE_forward restated (models/orn/src/cpu/RotationInvariantEncoding_cpu.cpp:6-45): feature [B,C,1,1] ->
    (mainDirection uint8 [B,C/nOri], aligned): first index of the strict maximum of each group, group rotated so that it
    comes first.  numpy (integer / copy work)."""
    f = _f32(feature)
    B, C = f.shape[:2]
    g = f.reshape(B, C // n_ori, n_ori)
    d = np.argmax(g, axis=2).astype(np.uint8)                     # argmax = first maximum = the `>` loop
    l = np.arange(n_ori)[None, None, :]
    src = (l + d[..., None]) % n_ori                              # aligned[a] = src[(a + d) % n]
    return d, np.take_along_axis(g, src, axis=2).reshape(f.shape)


def rie_backward(main_direction, grad_output, n_ori):
    """RIE_backward restated (:47-76): gradInput[(l + d) % n] = gradOutput[l]"""
    g = _f32(grad_output)
    d = np.asarray(main_direction, np.int64)
    B, F = d.shape
    gg = g.reshape(B, F, n_ori)
    a = np.arange(n_ori)[None, None, :]
    src = (a - d[..., None]) % n_ori                              # gradInput[a] = gradOutput[(a - d) % n]
    return np.take_along_axis(gg, src, axis=2).reshape(g.shape)


def rot_inv_pool(x, n_ori=8):
    x = _f32(x)
    B, C, H, W = x.shape
    out = np.empty((B, C // n_ori, H, W), np.float32)
    lib().orc_rot_inv_pool(_p(x), B, C, H * W, n_ori, _p(out))
    return out


def deform_conv_forward(x, offset, weight, stride=(1, 1), padding=(1, 1), dilation=(1, 1),
                        groups=1, deformable_groups=1, f16_cols=False, relu=False):
    x, offset, weight = _f32(x), _f32(offset), _f32(weight)
    B, C, H, W = x.shape
    O, _, kH, kW = weight.shape
    Ho = (H + 2 * padding[0] - (dilation[0] * (kH - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (kW - 1) + 1)) // stride[1] + 1
    assert offset.shape == (B, deformable_groups * 2 * kH * kW, Ho, Wo), offset.shape
    out = np.empty((B, O, Ho, Wo), np.float32)
    lib().orc_deform_conv_forward(_p(x), _p(offset), _p(weight), B, C, H, W, O, kH, kW,
                                  stride[0], stride[1], padding[0], padding[1],
                                  dilation[0], dilation[1], groups, deformable_groups,
                                  int(f16_cols), int(relu), _p(out))
    return out


def deform_conv_forward_f64(x, offset, weight, stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1,
                            deformable_groups=1):
    """the reference's scalar_t = double instantiation (AT_DISPATCH_FLOATING_TYPES_AND_HALF, deform_conv_cuda_kernel.cu:258):
    deformable_im2col (:189-242) + deformable_im2col_bilinear (:83-114) + the per-group GEMM (deform_conv_cuda.cpp:222-237),
    every operation in float64 -- plain numpy, small shapes only (test infrastructure like the rest of this package)"""
    x, offset, weight = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, offset, weight))
    B, C, H, W = x.shape
    O, Cg, kH, kW = weight.shape
    Ho = (H + 2 * padding[0] - (dilation[0] * (kH - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (kW - 1) + 1)) // stride[1] + 1
    assert offset.shape == (B, deformable_groups * 2 * kH * kW, Ho, Wo) and Cg * groups == C
    cols = np.zeros((B, C, kH * kW, Ho, Wo), np.float64)
    ho, wo = np.meshgrid(np.arange(Ho), np.arange(Wo), indexing="ij")
    cpdg = C // deformable_groups
    for b in range(B):
        for c in range(C):
            dg = c // cpdg
            im = x[b, c]
            for i in range(kH):
                for j in range(kW):
                    t = i * kW + j
                    h_im = (ho * stride[0] - padding[0] + i * dilation[0]) + offset[b, dg * 2 * kH * kW + 2 * t]
                    w_im = (wo * stride[1] - padding[1] + j * dilation[1]) + offset[b, dg * 2 * kH * kW + 2 * t + 1]
                    ok = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)            # :228
                    hl, wl = np.floor(h_im).astype(np.int64), np.floor(w_im).astype(np.int64)
                    hh_, wh_ = hl + 1, wl + 1
                    lh, lw = h_im - hl, w_im - wl
                    hh, hw = 1 - lh, 1 - lw

                    def at(yy, xx, keep):
                        keep = keep & ok
                        return np.where(keep, im[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)
                    v1 = at(hl, wl, (hl >= 0) & (wl >= 0))                              # :97-108
                    v2 = at(hl, wh_, (hl >= 0) & (wh_ <= W - 1))
                    v3 = at(hh_, wl, (hh_ <= H - 1) & (wl >= 0))
                    v4 = at(hh_, wh_, (hh_ <= H - 1) & (wh_ <= W - 1))
                    cols[b, c, t] = np.where(ok, hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4, 0.0)   # :110-112
    out = np.empty((B, O, Ho, Wo), np.float64)
    Og = O // groups
    for g in range(groups):
        out[:, g * Og:(g + 1) * Og] = np.einsum("okt,bkthw->bohw", weight[g * Og:(g + 1) * Og].reshape(Og, Cg, kH * kW),
                                                cols[:, g * Cg:(g + 1) * Cg])
    return out


def deform_conv_forward_half(x, offset, weight, stride=(1, 1), padding=(1, 1), dilation=(1, 1), relu=False):
    """the reference's scalar_t = Half instantiation (models/dcn/deform_conv.py:45-46 casts the offsets to the input's
    dtype; deform_conv_cuda_kernel.cu:83-114,221-228 then compute h_im / w_im, the bilinear weights and the blend in
    c10::Half, every operation rounded to binary16).  x, weight: binary16-representable float32; offset: float32 (rounded
    inside).  groups = deformable_groups = 1.  -> out float32 holding binary16 values"""
    x, offset, weight = _f32(x), _f32(offset), _f32(weight)
    B, C, H, W = x.shape
    O, _, kH, kW = weight.shape
    Ho = (H + 2 * padding[0] - (dilation[0] * (kH - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (kW - 1) + 1)) // stride[1] + 1
    assert offset.shape == (B, 2 * kH * kW, Ho, Wo), offset.shape
    assert np.array_equal(x, x.astype(np.float16).astype(np.float32)), "x must be binary16-representable"
    out = np.empty((B, O, Ho, Wo), np.float32)
    lib().orc_deform_conv_forward_half(_p(x), _p(offset), _p(weight), B, C, H, W, O, kH, kW, stride[0], stride[1],
                                       padding[0], padding[1], dilation[0], dilation[1], int(relu), _p(out))
    return out


# --------------------------------------------------------------------------- ORN index table
# ORConv2d.get_indices (models/orn/modules/ORConv.py:41-75): 1-based, [nOri,kH,kW,nRot] uint8.
_ROT3 = {0: (1, 2, 3, 4, 5, 6, 7, 8, 9), 45: (2, 3, 6, 1, 5, 9, 4, 7, 8),
         90: (3, 6, 9, 2, 5, 8, 1, 4, 7), 135: (6, 9, 8, 3, 5, 7, 2, 1, 4),
         180: (9, 8, 7, 6, 5, 4, 3, 2, 1), 225: (8, 7, 4, 9, 5, 1, 6, 3, 2),
         270: (7, 4, 1, 8, 5, 2, 9, 6, 3), 315: (4, 1, 2, 7, 5, 3, 8, 9, 6)}


def arf_indices(n_ori, n_rot, k=3):
    tab = _ROT3 if k == 3 else {a: (1,) for a in range(0, 360, 45)}
    idx = np.zeros((n_ori * k * k, n_rot), np.uint8)
    d_ori, d_rot = 360 / n_ori, 360 / n_rot
    for i in range(n_ori):
        for j in range(k * k):
            for r in range(n_rot):
                ang = d_rot * r
                layer = (i + int(np.floor(ang / d_ori))) % n_ori
                idx[i * k * k + j, r] = layer * k * k + tab[int(ang)][j]
    return idx.reshape(n_ori, k, k, n_rot)


# --------------------------------------------------------------------------- head glue (numpy f32)
def grid_anchors(feat_h, feat_w, stride, scale=4.0):
    """AnchorGeneratorRotated.gen_grid_anchors, 1 square anchor/position, angle 0
    (models/anchors.py:36-61, :75-126): centre = idx*stride + 0.5*(stride-1), side = scale*stride.
    Returns [H*W, 5] float32 (x, y, w, h, a), row-major over (y, x)."""
    xs = np.arange(feat_w, dtype=np.float32) * np.float32(stride) + np.float32(0.5 * (stride - 1))
    ys = np.arange(feat_h, dtype=np.float32) * np.float32(stride) + np.float32(0.5 * (stride - 1))
    out = np.zeros((feat_h, feat_w, 5), np.float32)
    out[..., 0] = xs[None, :]
    out[..., 1] = ys[:, None]
    out[..., 2] = np.float32(scale * stride)
    out[..., 3] = np.float32(scale * stride)
    return out.reshape(-1, 5)


def norm_angle(a):
    """utils/general.py:925-929: (a + pi/4) mod pi - pi/4 (python/torch floor-mod)."""
    a = np.asarray(a, np.float32)
    lo = np.float32(-np.pi / 4)
    return (np.mod(a - lo, np.float32(np.pi)) + lo).astype(np.float32)


def delta2bbox_rotated(rois, deltas, wh_ratio_clip=16 / 1000):
    """models/boxes.py:82-162 (is_encode_relative=True).  float32 throughout."""
    r, d = _f32(rois), _f32(deltas)
    max_ratio = np.float32(np.abs(np.log(wh_ratio_clip)))
    dx, dy = d[:, 0], d[:, 1]
    dw = np.clip(d[:, 2], -max_ratio, max_ratio)
    dh = np.clip(d[:, 3], -max_ratio, max_ratio)
    ca, sa = np.cos(r[:, 4]), np.sin(r[:, 4])
    gx = dx * r[:, 2] * ca - dy * r[:, 3] * sa + r[:, 0]
    gy = dx * r[:, 2] * sa + dy * r[:, 3] * ca + r[:, 1]
    gw = r[:, 2] * np.exp(dw)
    gh = r[:, 3] * np.exp(dh)
    ga = norm_angle(np.float32(np.pi) * d[:, 4] + r[:, 4])
    return np.stack([gx, gy, gw, gh, ga], -1).astype(np.float32)


def deform_conv_backward(x, offset, weight, grad_out, stride=(1, 1), padding=(1, 1), dilation=(1, 1),
                         deformable_groups=1):
    """deform_conv_backward_input_cuda + _parameters_cuda restated (groups = 1):
    -> (grad_input, grad_offset, grad_weight), float32"""
    x, offset, weight, grad_out = _f32(x), _f32(offset), _f32(weight), _f32(grad_out)
    B, C, H, W = x.shape
    O, _, kH, kW = weight.shape
    gx, goff, gw = np.zeros_like(x), np.zeros_like(offset), np.zeros_like(weight)
    lib().orc_deform_conv_backward(_p(x), _p(offset), _p(weight), _p(grad_out), B, C, H, W, O, kH, kW,
                                   stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
                                   deformable_groups, _p(gx), _p(goff), _p(gw))
    return gx, goff, gw


def align_offsets(anchors, feat_h, feat_w, stride, k=3):
    """AlignConv.get_offset (models/alignconv.py:30-87) for ONE image.
    anchors [H*W,5] px/rad -> [2*k*k, H, W]; channel 2t = dy, 2t+1 = dx, tap t = ky*k+kx."""
    a = _f32(anchors)
    pad = (k - 1) // 2
    idx = np.arange(-pad, pad + 1, dtype=np.float32)
    yy, xx = np.meshgrid(idx, idx, indexing="ij")
    xx, yy = xx.reshape(-1), yy.reshape(-1)
    xc = np.arange(feat_w, dtype=np.float32)
    yc = np.arange(feat_h, dtype=np.float32)
    ycg, xcg = np.meshgrid(yc, xc, indexing="ij")
    x_conv = xcg.reshape(-1)[:, None] + xx
    y_conv = ycg.reshape(-1)[:, None] + yy
    s = np.float32(stride)
    x_ctr, y_ctr, w, h = a[:, 0] / s, a[:, 1] / s, a[:, 2] / s, a[:, 3] / s
    cos, sin = np.cos(a[:, 4]), np.sin(a[:, 4])
    dw, dh = w / np.float32(k), h / np.float32(k)
    x, y = dw[:, None] * xx, dh[:, None] * yy
    xr = cos[:, None] * x - sin[:, None] * y
    yr = sin[:, None] * x + cos[:, None] * y
    off_x = xr + x_ctr[:, None] - x_conv
    off_y = yr + y_ctr[:, None] - y_conv
    off = np.stack([off_y, off_x], -1).reshape(a.shape[0], -1)
    return np.ascontiguousarray(off.T.reshape(-1, feat_h, feat_w), np.float32)


def multiclass_nms_rotated(bboxes, scores, score_thr=0.05, iou_thr=0.5, max_per_img=2000,
                           rule=RULE_GT, sort_mode=SORT_GPU):
    """utils/bbox_nms_rotated.py:5-64.  Returns ([K,6] x,y,w,h,a,score ; labels float32 [K])."""
    b, s = _f32(bboxes), _f32(scores)
    mask = s > np.float32(score_thr)
    rows, cols = np.nonzero(mask)          # row-major, same order as torch boolean indexing
    cb, cs, cl = b[rows], s[rows, cols], cols.astype(np.float32)
    if cb.shape[0] == 0:
        return np.zeros((0, 6), np.float32), np.zeros((0,), np.float32)
    keep = ml_nms_rotated(cb, cs, cl, iou_thr, rule=rule, sort_mode=sort_mode)
    cb, cs, cl = cb[keep], cs[keep], cl[keep]
    if keep.shape[0] > max_per_img:
        inds = np.argsort(-cs, kind="stable")[:max_per_img]
        cb, cs, cl = cb[inds], cs[inds], cl[inds]
    return np.concatenate([cb, cs[:, None]], 1), cl


def rboxes_to_polys(rboxes):
    """corner points of (x,y,w,h,a) boxes in double, order as get_rotated_vertices
    (box_iou_rotated_utils.h:56-75) — used to feed polyiou the same rectangles."""
    r = np.asarray(rboxes, np.float64).reshape(-1, 5)
    c, s = np.cos(r[:, 4]) * 0.5, np.sin(r[:, 4]) * 0.5
    x, y, w, h = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    p0x, p0y = x - s * h - c * w, y + c * h - s * w
    p1x, p1y = x + s * h - c * w, y - c * h - s * w
    return np.stack([p0x, p0y, p1x, p1y, 2 * x - p0x, 2 * y - p0y, 2 * x - p1x, 2 * y - p1y], 1)
