"""Chip-merge of DOTA Task-1 result files with the polygon NMS on the GPU.

``mergesingle`` / ``mergebypoly`` follow DOTA_devkit/ResultMerge_multi_process.py:180-243, :279-296: every line
``<chip name> <confidence> x1 y1 ... x4 y4`` is moved back to its original image (chip name
``<image>__<rate>__<x>___<y>``, :199-213; ``poly2origpoly`` :171-178), the detections of one image go through
``py_cpu_nms_poly_fast`` -- here ``s2anet_amd.rotated.nms_poly`` (s2a_nms_poly: HBB cull, polyiou, greedy scan,
all on the device) -- and the survivors are written ``<image> <confidence> <8 coordinates>`` with Python's
``str(float)`` exactly as the script does.
"""
import os
import re

import numpy as np
import torch

from .rotated import nms_poly

_XY = re.compile(r"__\d+___\d+")
_RATE = re.compile(r"__([\d+\.]+)__\d+___")


def parse_chip_name(subname):
    """'P0003__1__824___1648' -> ('P0003', 824, 1648, '1')  (:199-213)"""
    oriname = subname.split("__")[0]
    x, y = (int(v) for v in re.findall(r"\d+", _XY.findall(subname)[0])[:2])
    return oriname, x, y, _RATE.findall(subname)[0]


def poly2origpoly(poly, x, y, rate):
    """:171-178, vectorised: (coordinate + chip offset) / rate in double"""
    p = np.asarray(poly, np.float64).reshape(-1, 8).copy()
    p[:, 0::2] = (p[:, 0::2] + np.asarray(x, np.float64).reshape(-1, 1)) / np.asarray(rate, np.float64).reshape(-1, 1)
    p[:, 1::2] = (p[:, 1::2] + np.asarray(y, np.float64).reshape(-1, 1)) / np.asarray(rate, np.float64).reshape(-1, 1)
    return p


def merge_lines(lines, thresh=0.5, device="cuda"):
    """the body of mergesingle on a list of result lines -> list of merged output lines (no newline)"""
    names, dets = {}, []
    for ln in lines:
        sp = ln.strip().split(" ")
        if len(sp) < 10:
            continue
        ori, x, y, rate = parse_chip_name(sp[0])
        names.setdefault(ori, []).append(len(dets))
        dets.append((x, y, float(rate), float(sp[1])) + tuple(map(float, sp[2:10])))
    if not dets:
        return []
    a = np.asarray(dets, np.float64)
    polys = poly2origpoly(a[:, 4:12], a[:, 0], a[:, 1], a[:, 2])
    d9 = np.concatenate([polys, a[:, 3:4]], 1)
    out = []
    for ori, idx in names.items():                       # insertion order = the script's dict order
        sub = d9[idx]
        keep = nms_poly(torch.from_numpy(sub).to(device), thresh).cpu().numpy()
        for k in keep:
            det = sub[k].tolist()
            out.append(ori + " " + str(det[-1]) + " " + " ".join(map(str, det[:-1])))
    return out


def mergesingle(dstpath, fullname, thresh=0.5, device="cuda"):
    """mergesingle(dstpath, nms, fullname) with nms = py_cpu_nms_poly_fast(thresh) on the GPU"""
    name = os.path.basename(os.path.splitext(fullname)[0])
    with open(fullname, "r") as f:
        out = merge_lines(f.readlines(), thresh, device)
    with open(os.path.join(dstpath, name + ".txt"), "w") as f:
        for ln in out:
            f.write(ln + "\n")


def mergebypoly(srcpath, dstpath, thresh=0.5, device="cuda"):
    """:279-296: every class file of srcpath"""
    for root, _, files in os.walk(srcpath):
        for fn in files:
            mergesingle(dstpath, os.path.join(root, fn), thresh, device)
