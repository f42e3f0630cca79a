"""Output / wire formats of the detection path: val.py:40-52, utils/general.py:629-648,
:886-921.  Detections stay on the device until the text is produced."""
import os.path as osp

import torch

from . import _lib


def scale_coords_rotated(img1_shape, bboxes, img0_shape, ratio_pad=None):
    """utils/general.py:629-648: network-input coordinates -> original-image coordinates, in place.
    bboxes[N,>=5] (x, y, w, h, theta, ...)"""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    bboxes[:, 0] -= pad[0]
    bboxes[:, 1] -= pad[1]
    bboxes[:, :4] /= gain
    return bboxes


def rbox_to_poly(boxes):
    """rotated_box_to_poly_single (utils/general.py:886-921, cv2.boxPoints) for all rows at once:
    boxes[N,>=5] f32 on the GPU -> polys[N,8] f32"""
    _lib.require_cuda(boxes)
    b = boxes.float()
    if b.stride(-1) != 1:
        b = b.contiguous()
    n = b.shape[0]
    out = torch.empty((n, 8), dtype=torch.float32, device=b.device)
    if n:
        with torch.cuda.device(b.device):
            _lib.check(_lib.lib().s2a_rbox_to_poly(_lib.ptr(b), n, b.stride(0), _lib.ptr(out), _lib.stream_ptr(b.device)))
    return out


def task1_lines(img_name, det_bboxes, det_labels, class_names):
    """val.py:40-52: one image's detections [K,6] (x,y,w,h,theta,score) + labels -> {class name: [lines]},
    '<image> <score:.4f> <x1:.4f> ... <y4:.4f>\\n'"""
    polys = rbox_to_poly(det_bboxes[:, :5]).cpu().tolist()
    scores = det_bboxes[:, 5].float().cpu().tolist()
    labels = det_labels.long().cpu().tolist()
    stem = osp.splitext(img_name)[0]
    out = {}
    for p, sc, lb in zip(polys, scores, labels):
        out.setdefault(class_names[lb], []).append(
            "{} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f}\n".format(stem, sc, *p))
    return out
