#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the REFERENCE itself.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
  * native ops  : the reference's CPU sources built unmodified into oracle/_ref
                  (oracle/build_ref.py) + its geometry header host-compiled with __CUDACC__
  * head glue   : the reference's Python modules imported in place (sys.path), with stub
                  modules for the absent third-party cv2 / torchvision and the oracle/_ref
                  CPU builds standing in for the *_cuda extension modules
The fixtures hold DATA only (inputs + expected outputs).  The reference never travels to the
GPU box; these arrays do.
"""
import collections
import ctypes
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

from oracle import build_ref, ref  # noqa: E402
import oracle  # noqa: E402

torch.set_num_threads(1)
SEED = 1234


def rand_boxes(rng, n, span=1024.0, lo=4.0, hi=100.0):
    b = np.empty((n, 5), np.float32)
    b[:, 0:2] = rng.uniform(0, span, (n, 2))
    b[:, 2:4] = rng.uniform(lo, hi, (n, 2))
    b[:, 4] = rng.uniform(-np.pi / 4, 3 * np.pi / 4, n)
    return b


EDGE_BOXES = np.array([
    [50, 50, 20, 10, 0.0],            # base
    [50, 50, 20, 10, 0.0],            # identical
    [50, 50, 10, 5, 0.0],             # contained
    [70, 50, 20, 10, 0.0],            # shared edge
    [50, 60, 20, 10, 0.0],            # shared long edge
    [50, 50, 20, 10, np.pi / 4],      # 45 deg
    [50, 50, 10, 10, np.pi / 4],      # square in square rotated (0.707 case scale)
    [50, 50, 20, 10, np.pi / 2],      # 90 deg
    [50, 50, 20, 10, -np.pi / 4],     # angle range ends
    [50, 50, 20, 10, 3 * np.pi / 4],
    [50, 50, 1e-8, 1e-8, 0.3],        # area < 1e-14
    [50, 50, 0, 10, 0.0],             # zero width
    [500, 500, 20, 10, 0.2],          # far away
    [55, 52, 18, 9, 0.1],             # generic overlap
    [0.5, 0.5, 1, 1, 0],              # the reference's own 1/7 example
    [1.0, 1.0, 1, 1, 0],
    [50, 50, 20, 10, 1e-4],           # nearly parallel
    [60, 50, 20, 10, 1e-7],
    [50.000004, 50, 20, 10, 0.0],     # sub-ulp shifts
    [1e4, 1e4, 300, 200, 0.7],        # large coordinates
    [1e4 + 10, 1e4 - 5, 250, 220, -0.3],
], np.float32)


def gen_iou(rng):
    b1 = np.concatenate([EDGE_BOXES, rand_boxes(rng, 256 - len(EDGE_BOXES), span=300)])
    b2 = np.concatenate([EDGE_BOXES[::-1], rand_boxes(rng, 256 - len(EDGE_BOXES), span=300)])
    iou_cpu = ref.box_iou_rotated()(torch.from_numpy(b1), torch.from_numpy(b2)).numpy()
    # GPU (swap-sort) branch of the reference header, all pairs
    L = ref.geom_gpubranch()
    f32p = ctypes.POINTER(ctypes.c_float)
    n, m = b1.shape[0], b2.shape[0]
    a6 = np.zeros((n * m, 6), np.float32)
    c6 = np.zeros((n * m, 6), np.float32)
    a6[:, :5] = np.repeat(b1, m, 0)
    c6[:, :5] = np.tile(b2, (n, 1))
    iou_gpu = np.empty(n * m, np.float32)
    L.ref_gpubranch_iou6_pairs(a6.ctypes.data_as(f32p), c6.ctypes.data_as(f32p), n * m,
                               iou_gpu.ctypes.data_as(f32p))
    iou_gpu = iou_gpu.reshape(n, m)
    # polyiou (double) on the overlapping pairs
    P1, P2 = oracle.rboxes_to_polys(b1), oracle.rboxes_to_polys(b2)
    ii, jj = np.nonzero(iou_cpu > 0)
    sel = np.arange(len(ii))[:4000]
    fpoly = ref.polyiou()
    poly = np.array([fpoly(P1[ii[k]], P2[jj[k]]) for k in sel], np.float64)
    np.savez_compressed(os.path.join(OUT, "iou_256.npz"), boxes1=b1, boxes2=b2,
                        iou_ref_cpu=iou_cpu, iou_ref_gpubranch=iou_gpu,
                        poly_i=ii[sel].astype(np.int32), poly_j=jj[sel].astype(np.int32),
                        poly_iou=poly)
    print("iou_256: nonzero %.3f  cpu!=gpu-branch: %d" % (
        (iou_cpu > 0).mean(), (iou_cpu.view(np.uint32) != iou_gpu.view(np.uint32)).sum()))


def gen_nms(rng):
    n = 2048
    d = rand_boxes(rng, n, span=320)
    s = (rng.permutation(n).astype(np.float32) + 1) / np.float32(n + 1) * np.float32(0.95) + np.float32(0.05)
    assert len(np.unique(s)) == n
    lab = rng.integers(0, 15, n).astype(np.float32)
    td, ts, tl = torch.from_numpy(d), torch.from_numpy(s), torch.from_numpy(lab)
    out = dict(dets=d, scores=s, labels=lab)
    for thr in (0.1, 0.5):
        k_ml = ref.ml_nms_rotated()(td, ts, tl, thr).numpy()
        k_sc = ref.nms_rotated()(td, ts, thr).numpy()
        out[f"ml_keep_ge_{thr}"] = k_ml
        out[f"sc_keep_ge_{thr}"] = k_sc
        # GPU rule (>) with GPU sort branch: restated (reference CUDA cannot run here)
        out[f"ml_keep_gt_{thr}"] = oracle.ml_nms_rotated(d, s, lab, thr, rule=oracle.RULE_GT,
                                                         sort_mode=oracle.SORT_GPU)
        out[f"sc_keep_gt_{thr}"] = oracle.nms_rotated(d, s, thr, rule=oracle.RULE_GT,
                                                      sort_mode=oracle.SORT_GPU)
        out[f"ml_margin_{thr}"] = np.float64(oracle.nms_margin(d, lab, thr))
        out[f"sc_margin_{thr}"] = np.float64(oracle.nms_margin(d, None, thr))
        print("nms thr", thr, "ml keep", len(k_ml), "sc keep", len(k_sc),
              "margins", out[f"ml_margin_{thr}"], out[f"sc_margin_{thr}"])
    # the reference's 4-box sanity case
    d4 = np.array([[10, 10, 8, 4, 0.3], [10, 10, 8, 4, 0.3], [10, 10, 8, 4, 0.3], [100, 100, 8, 4, 0]], np.float32)
    s4 = np.array([0.9, 0.8, 0.7, 0.6], np.float32)
    l4 = np.array([0, 0, 1, 0], np.float32)
    out["d4"], out["s4"], out["l4"] = d4, s4, l4
    out["ml_keep4"] = ref.ml_nms_rotated()(torch.from_numpy(d4), torch.from_numpy(s4), torch.from_numpy(l4), 0.5).numpy()
    out["sc_keep4"] = ref.nms_rotated()(torch.from_numpy(d4), torch.from_numpy(s4), 0.5).numpy()
    np.savez_compressed(os.path.join(OUT, "nms_2k.npz"), **out)


def gen_arf(rng):
    orn = ref.orn()
    out = {}
    for tag, shp in (("s1", (4, 2, 1, 3, 3)), ("s8", (4, 2, 8, 3, 3))):
        w = rng.standard_normal(shp).astype(np.float32)
        idx = oracle.arf_indices(shp[2], 8, 3)
        out[f"w_{tag}"] = w
        out[f"idx_{tag}"] = idx
        out[f"out_{tag}"] = orn.arf_forward(torch.from_numpy(w), torch.from_numpy(idx)).numpy()
    np.savez_compressed(os.path.join(OUT, "arf_small.npz"), **out)


def gen_arf_backward():
    """separate file + own rng so the other fixtures stay byte-identical"""
    rng = np.random.default_rng(4321)
    orn = ref.orn()
    out = {}
    for tag, shp in (("s1", (4, 2, 1, 3, 3)), ("s8", (4, 2, 8, 3, 3))):
        O, I, nOri, kH, kW = shp
        idx = oracle.arf_indices(nOri, 8, 3)
        g = rng.standard_normal((O * 8, I * nOri, kH, kW)).astype(np.float32)
        out[f"idx_{tag}"] = idx
        out[f"gout_{tag}"] = g
        out[f"gin_{tag}"] = orn.arf_backward(torch.from_numpy(idx), torch.from_numpy(g)).numpy()
    np.savez_compressed(os.path.join(OUT, "arf_backward_small.npz"), **out)


def gen_rie():
    """RotationInvariantEncoding forward / backward from the reference's CPU op (own rng: other fixtures unchanged)"""
    rng = np.random.default_rng(8642)
    orn = ref.orn()
    out = {}
    for tag, (B, F, n) in (("a", (3, 5, 8)), ("b", (2, 7, 4))):
        f = rng.standard_normal((B, F * n, 1, 1)).astype(np.float32)
        f[0, :n, 0, 0] = 0.25                                    # a tie: the first index wins
        f[1, n:2 * n, 0, 0] = np.float32(-1e30)                  # all equal and very negative
        d, al = orn.rie_forward(torch.from_numpy(f), n)
        g = rng.standard_normal((B, F * n, 1, 1)).astype(np.float32)
        gi = orn.rie_backward(d, torch.from_numpy(g), n)
        out[f"f_{tag}"], out[f"n_{tag}"] = f, np.int64(n)
        out[f"dir_{tag}"], out[f"aligned_{tag}"] = d.numpy(), al.numpy()
        out[f"gout_{tag}"], out[f"gin_{tag}"] = g, gi.numpy()
    np.savez_compressed(os.path.join(OUT, "rie_small.npz"), **out)


def gen_merge_nms():
    """chip-merge polygon NMS from the reference's own script (DOTA_devkit/ResultMerge_multi_process.py),
    imported in place with the SWIG polyiou module built into oracle/_ref and a stub for shapely"""
    rng = np.random.default_rng(999)
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref", "polyiou"))
    sh = types.ModuleType("shapely")
    sh.geometry = types.ModuleType("shapely.geometry")
    sys.modules.setdefault("shapely", sh)
    sys.modules.setdefault("shapely.geometry", sh.geometry)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from DOTA_devkit.ResultMerge_multi_process import py_cpu_nms_poly_fast
    n = 900
    polys = oracle.rboxes_to_polys(rand_boxes(rng, n, span=420))
    scores = (rng.permutation(n) + 1.0) / (n + 1.0)
    dets = np.concatenate([polys, scores[:, None]], 1).astype(np.float64)
    out = {"dets": dets}
    for thr in (0.1, 0.5):
        out[f"keep_{thr}"] = np.asarray(py_cpu_nms_poly_fast(dets, thr), np.int64)
        print("merge nms thr", thr, "keep", len(out[f"keep_{thr}"]))
    np.savez_compressed(os.path.join(OUT, "merge_nms_poly.npz"), **out)


def import_reference_python():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    cv2 = types.ModuleType("cv2")
    cv2.setNumThreads = lambda *a, **k: None
    sys.modules.setdefault("cv2", cv2)
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    from oracle.ref import _load_pyext
    sys.modules["utils.box_iou_rotated.box_iou_rotated_cuda"] = _load_pyext("ref_box_iou_rotated")
    sys.modules["utils.ml_nms_rotated.ml_nms_rotated_cuda"] = _load_pyext("ref_ml_nms_rotated")
    sys.modules["utils.nms_rotated.nms_rotated_cuda"] = _load_pyext("ref_nms_rotated")
    sys.modules["models.orn.orn_cuda"] = _load_pyext("ref_orn")
    sys.modules["models.dcn.deform_conv_cuda"] = types.ModuleType("deform_conv_cuda")
    sys.modules["models.dcn.deform_pool_cuda"] = types.ModuleType("deform_pool_cuda")


def gen_glue(rng):
    import_reference_python()
    from models.anchors import AnchorGeneratorRotated
    from models.alignconv import AlignConv
    from models.boxes import rboxes_decode
    from models.orn import ORConv2d
    from utils.bbox_nms_rotated import multiclass_nms_rotated
    g = torch.Generator().manual_seed(SEED)
    out = {}
    # grid anchors, two levels
    for stride, (h, w) in ((8, (12, 20)), (32, (5, 7))):
        ag = AnchorGeneratorRotated(stride, [4], [1.0], angles=[0])
        out[f"anchors_s{stride}"] = ag.gen_grid_anchors((h, w), stride).reshape(-1, 5).numpy()
    # decode with both clips (head.py:48 uses 1e-6; boxes.py:226 default 16/1000)
    anc = torch.from_numpy(out["anchors_s8"]).clone()
    anc[:, 4] = (torch.rand(anc.shape[0], generator=g) - 0.25) * np.pi
    deltas = torch.randn(anc.shape[0], 5, generator=g) * torch.tensor([0.5, 0.5, 1.5, 1.5, 0.3])
    deltas[0, 2] = 20.0   # exercise both clamps
    deltas[1, 3] = -20.0
    out["dec_anchors"], out["dec_deltas"] = anc.numpy(), deltas.numpy()
    out["dec_clip_fam"] = rboxes_decode(anc, deltas, wh_ratio_clip=1e-6).numpy()
    out["dec_clip_odm"] = rboxes_decode(anc, deltas).numpy()
    # AlignConv.get_offset on refined (rotated) anchors, stride 8, 12x20 map
    ac = AlignConv(8, 8, kernel_size=3)
    refined = torch.from_numpy(out["dec_clip_fam"]).clone()
    refined[:, 2:4] = refined[:, 2:4].clamp(max=400.0)
    out["off_anchors"] = refined.numpy()
    out["off_s8"] = ac.get_offset(refined, (12, 20), 8).numpy()
    # ORConv2d index table
    oc = ORConv2d(16, 2, kernel_size=3, padding=1, arf_config=(1, 8))
    out["orconv_indices_1_8"] = oc.indices.numpy()
    oc8 = ORConv2d(16, 2, kernel_size=3, padding=1, arf_config=(8, 8))
    out["orconv_indices_8_8"] = oc8.indices.numpy()
    # multiclass_nms_rotated end to end (reference python + reference CPU ml_nms => ">=" rule)
    nb = 600
    bb = torch.from_numpy(rand_boxes(rng, nb, span=220))
    sc = torch.rand(nb, 15, generator=g) ** 6      # sparse-ish scores
    det, lab = multiclass_nms_rotated(bb, sc, score_thr=0.05, iou_thr=0.5, max_per_img=300)
    out["mc_bboxes"], out["mc_scores"] = bb.numpy(), sc.numpy()
    out["mc_det"], out["mc_labels"] = det.numpy(), lab.numpy()
    det0, lab0 = multiclass_nms_rotated(bb, sc * 0, score_thr=0.05, iou_thr=0.5, max_per_img=300)
    out["mc_empty_det_shape"] = np.array(det0.shape)
    out["mc_empty_lab_shape"] = np.array(lab0.shape)
    np.savez_compressed(os.path.join(OUT, "head_glue.npz"), **out)
    print("glue:", {k: v.shape for k, v in out.items()})


def gen_dcn(rng):
    """Deformable conv: the reference has NO runnable implementation here (CUDA only).
    Expected values come from an independent pure-torch formulation (explicit 4-corner
    gather written from deform_conv_cuda_kernel.cu:83-114,189-242 semantics) — recorded so
    the oracle restatement and the HIP kernel are checked against the same numbers."""
    g = torch.Generator().manual_seed(SEED)
    B, C, H, W, O = 2, 16, 9, 11, 8
    x = torch.randn(B, C, H, W, generator=g)
    wgt = torch.randn(O, C, 3, 3, generator=g) * 0.1
    off = torch.randn(B, 18, H, W, generator=g) * 2.0
    off[0, :, 0, 0] = 50.0      # far outside -> zeros
    off[0, :, 1, 1] = -0.999    # just inside the (-1, ...) border rule
    out = torch_deform_conv(x, off, wgt)
    np.savez_compressed(os.path.join(OUT, "dcn_small.npz"), x=x.numpy(), weight=wgt.numpy(),
                        offset=off.numpy(), out_torch=out.numpy())


def gen_voc_eval():
    """DOTA Task-1 evaluation of one class by the reference's own voc_eval (dota_evaluation_task1.py), run on
    synthetic detection / annotation files written to a scratch directory; the arrays and its rec / prec / ap are
    the fixture.  (numpy >= 1.24 dropped the np.bool alias the script uses: restored for the call.)"""
    import tempfile, shutil
    rng = np.random.default_rng(1357)
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF, "DOTA_devkit"))
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref", "polyiou"))
    if not hasattr(np, "bool"):
        np.bool = bool
    import dota_evaluation_task1 as ev
    n_img, cls = 12, "plane"
    gt_polys, gt_img, gt_diff, det_polys, det_img, det_sc = [], [], [], [], [], []
    for im in range(n_img):
        ng = int(rng.integers(0, 9))
        g = oracle.rboxes_to_polys(rand_boxes(rng, ng, span=600)) if ng else np.zeros((0, 8))
        for q in g:
            gt_polys.append(np.round(q, 1)); gt_img.append(im); gt_diff.append(int(rng.random() < 0.2))
        for q in g:                                   # detections near the ground truth, some duplicated ...
            for _ in range(int(rng.integers(0, 3))):
                det_polys.append(np.round(q + rng.normal(0, 2.5, 8), 1)); det_img.append(im); det_sc.append(rng.random())
        for q in oracle.rboxes_to_polys(rand_boxes(rng, int(rng.integers(0, 6)), span=600)):   # ... and false alarms
            det_polys.append(np.round(q, 1)); det_img.append(im); det_sc.append(rng.random() * 0.8)
    det_sc = np.round(np.array(det_sc) + np.arange(len(det_sc)) * 1e-6, 6)     # distinct confidences
    tmp = tempfile.mkdtemp(prefix="s2a_voc_")
    try:
        names = ["P%04d" % i for i in range(n_img)]
        open(os.path.join(tmp, "set.txt"), "w").write("\n".join(names) + "\n")
        for im, nm in enumerate(names):
            with open(os.path.join(tmp, nm + ".txt"), "w") as f:
                for q, gi, d in zip(gt_polys, gt_img, gt_diff):
                    if gi == im:
                        f.write(" ".join("%.1f" % v for v in q) + " %s %d\n" % (cls, d))
                f.write("1 2 3 4 5 6 7 8 ship 0\n")                              # another class: ignored
        with open(os.path.join(tmp, "Task1_%s.txt" % cls), "w") as f:
            for q, di, sc in zip(det_polys, det_img, det_sc):
                f.write("%s %.6f " % (names[di], sc) + " ".join("%.1f" % v for v in q) + "\n")
        out = {"det_polys": np.array(det_polys), "det_scores": det_sc, "det_image": np.array(det_img),
               "gt_polys": np.array(gt_polys), "gt_image": np.array(gt_img), "gt_difficult": np.array(gt_diff),
               "num_images": np.array(n_img)}
        for tag, kw in (("default", {}), ("voc07", dict(use_07_metric=True)), ("hard", dict(is_filter_difficult=False)),
                        ("thr07", dict(ovthresh=0.7))):
            rec, prec, ap, sc = ev.voc_eval(os.path.join(tmp, "Task1_{:s}.txt"), os.path.join(tmp, "{:s}.txt"),
                                            os.path.join(tmp, "set.txt"), cls, **kw)
            out["rec_" + tag], out["prec_" + tag], out["ap_" + tag] = rec, prec, np.array(ap)
            print("voc_eval", tag, "ap", float(ap), "dets", len(det_sc), "gts", len(gt_diff))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(OUT, "voc_eval.npz"), **out)


def gen_merge_file():
    """chip-merge of one class file by the reference's own mergesingle (ResultMerge_multi_process.py:180-243),
    run on a synthetic result file in a scratch directory; input and output LINES are the fixture"""
    import tempfile, shutil, warnings
    rng = np.random.default_rng(8642)
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref", "polyiou"))
    sh = types.ModuleType("shapely")
    sh.geometry = types.ModuleType("shapely.geometry")
    sys.modules.setdefault("shapely", sh)
    sys.modules.setdefault("shapely.geometry", sh.geometry)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from DOTA_devkit.ResultMerge_multi_process import mergesingle, py_cpu_nms_poly_fast
    lines = []
    conf = iter(rng.permutation(np.arange(500, 10000))[:2000] / 10000.0)      # distinct confidences (ties: see DESIGN 2)
    for img, rate in (("P0003", "1"), ("P0170", "0.5"), ("P0009", "1")):
        r = float(rate)
        objs = oracle.rboxes_to_polys(rand_boxes(rng, 120, span=1800))        # objects in original-image pixels
        for cx, cy in ((0, 0), (824, 0), (0, 824), (824, 824)):               # 1024-px chips, 200 px overlap
            for q in objs:
                c = q.reshape(4, 2).mean(0) * r
                if cx <= c[0] < cx + 1024 and cy <= c[1] < cy + 1024:         # every chip that sees the object reports it
                    chip = q.reshape(4, 2) * r - np.array([cx, cy]) + rng.normal(0, 1.5, (4, 2))
                    lines.append("%s__%s__%d___%d %.4f " % (img, rate, cx, cy, next(conf)) +
                                 " ".join("%.4f" % v for v in chip.reshape(-1)))
    order = rng.permutation(len(lines))
    lines = [lines[i] for i in order]
    tmp = tempfile.mkdtemp(prefix="s2a_merge_")
    try:
        src = os.path.join(tmp, "Task1_ship.txt")
        open(src, "w").write("\n".join(lines) + "\n")
        os.mkdir(os.path.join(tmp, "out"))
        mergesingle(os.path.join(tmp, "out"), py_cpu_nms_poly_fast, src)
        merged = open(os.path.join(tmp, "out", "Task1_ship.txt")).read().splitlines()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("merge file:", len(lines), "->", len(merged))
    np.savez_compressed(os.path.join(OUT, "merge_file.npz"), lines=np.array(lines), merged=np.array(merged))


def gen_assign_labels():
    """label assignment from the reference's own models/utils.py:assign_labels run on its CPU box_iou_rotated"""
    import_reference_python()
    from models.utils import assign_labels
    rng = np.random.default_rng(2468)
    anchors = rand_boxes(rng, 3000, span=1000)
    anchors[:40, 0] = rng.uniform(-30, 0, 40)            # invalid anchors (outside the image)
    anchors[40:60, 2] = 1100.0
    gts = rand_boxes(rng, 37, span=1000)
    gts[:8] = anchors[100:108]                           # exact matches (IoU 1) ...
    gts[8:12] = anchors[200:204] + np.array([3, -2, 1, 0.5, 0.02], np.float32)
    anchors[300] = anchors[301]                          # ... and two anchors tying for a gt's maximum
    gts[12] = anchors[300] + np.array([6, 6, 0, 0, 0], np.float32)
    out = {"anchors": anchors, "gts": gts}
    for tag, kw in (("default", {}), ("first", dict(gt_max_assign_all=False)), ("thr", dict(pos_iou_thr=0.3, neg_iou_thr=0.2, min_pos_iou_thr=0.1)),
                    ("nofilter", dict(filter_invalid_anchors=False))):
        out["assign_" + tag] = assign_labels(torch.from_numpy(anchors), torch.from_numpy(gts), **kw).numpy()
        print("assign", tag, np.bincount(np.clip(out["assign_" + tag] + 2, 0, 3)))
    out["assign_empty"] = assign_labels(torch.from_numpy(anchors), torch.zeros((0, 5))).numpy()
    np.savez_compressed(os.path.join(OUT, "assign_labels.npz"), **out)


def gen_dcn_backward():
    """gradients of the deformable convolution: autograd through the independent torch formulation
    (torch_deform_conv; floor() has zero gradient, so d/d(offset) is the derivative of the bilinear
    interpolation the reference's get_coordinate_weight computes) -- no runnable reference exists"""
    g = torch.Generator().manual_seed(4242)
    B, C, H, W, O = 2, 12, 8, 10, 6
    x = torch.randn(B, C, H, W, generator=g).requires_grad_(True)
    wgt = (torch.randn(O, C, 3, 3, generator=g) * 0.1).requires_grad_(True)
    off = torch.randn(B, 18, H, W, generator=g) * 1.5
    off[0, :, 0, 0] = 40.0       # far outside: zero gradients
    off[1, :, 2, 3] = -0.75      # inside the (-1, ...) border band: some corners dropped
    off = off.requires_grad_(True)
    gout = torch.randn(B, O, H, W, generator=g)
    out = torch_deform_conv(x, off, wgt)
    out.backward(gout)
    np.savez_compressed(os.path.join(OUT, "dcn_backward_small.npz"), x=x.detach().numpy(), weight=wgt.detach().numpy(),
                        offset=off.detach().numpy(), grad_out=gout.numpy(), grad_input=x.grad.numpy(),
                        grad_offset=off.grad.numpy(), grad_weight=wgt.grad.numpy())
    print("dcn backward golden:", float(x.grad.abs().max()), float(off.grad.abs().max()), float(wgt.grad.abs().max()))


def gen_nms_f64():
    """nms_rotated / ml_nms_rotated on float64 boxes: the reference's CPU ops dispatch on the dtype of `dets`
    (AT_DISPATCH_FLOATING_TYPES, nms_rotated_cpu.cpp:66 / :67) and evaluate single_box_iou_rotated<double>.  The data
    are built so that float32 arithmetic DECIDES DIFFERENTLY: next to 400 random boxes sit 300 partners shifted along x
    by the amount that puts their double IoU within 2e-8 of the threshold (bisection on the oracle's double IoU), on
    either side -- a float32 evaluation (relative error ~1e-7) lands on the wrong side for a good part of them."""
    rng = np.random.default_rng(86420)
    thr = 0.5
    base = rand_boxes(rng, 400, span=900).astype(np.float64)
    base[:, 2:4] = rng.uniform(20, 90, (400, 2))
    partners = []
    for i in range(300):
        a = base[i]
        lo, hi = 0.0, float(a[2])                 # shift 0 -> IoU 1, shift w -> IoU small
        target = thr + (2e-8 if i % 2 else -2e-8)
        for _ in range(80):
            mid = 0.5 * (lo + hi)
            b = a.copy(); b[0] += mid * np.cos(a[4]); b[1] += mid * np.sin(a[4])
            v = oracle.iou_pairs_f64(a[None], b[None], sort_mode=oracle.SORT_CPU)[0]
            if v > target:
                lo = mid
            else:
                hi = mid
        b = a.copy(); b[0] += lo * np.cos(a[4]); b[1] += lo * np.sin(a[4])
        partners.append(b)
    dets = np.concatenate([base, np.array(partners)])
    n = len(dets)
    scores = (rng.permutation(n) + 1.0) / (n + 1.0)
    scores[400:] = scores[:300] - 1e-9 * (1 + np.arange(300))       # a partner right behind its box
    labels = np.concatenate([rng.integers(0, 5, 400), np.zeros(300)]).astype(np.float64)
    labels[400:] = labels[:300]
    td, ts, tl = torch.from_numpy(dets), torch.from_numpy(scores), torch.from_numpy(labels)
    out = dict(dets=dets, scores=scores, labels=labels, thr=np.float64(thr))
    out["ml_keep_f64"] = ref.ml_nms_rotated()(td, ts, tl, thr).numpy()
    out["sc_keep_f64"] = ref.nms_rotated()(td, ts, thr).numpy()
    out["ml_keep_f32"] = ref.ml_nms_rotated()(td.float(), ts.float(), tl.float(), thr).numpy()
    print("nms f64: n", n, "keep f64", len(out["ml_keep_f64"]), "keep f32", len(out["ml_keep_f32"]),
          "f64 != f32 decisions:", len(set(out["ml_keep_f64"]) ^ set(out["ml_keep_f32"])))
    np.savez_compressed(os.path.join(OUT, "nms_f64.npz"), **out)


def gen_formats():
    """scale_coords_rotated from the reference's own utils/general.py:629-648 (imported with the cv2 stub): letterboxed
    network coordinates -> original-image coordinates, with and without an explicit ratio_pad"""
    import_reference_python()
    from utils.general import scale_coords_rotated
    rng = np.random.default_rng(97531)
    d = np.concatenate([rand_boxes(rng, 64, span=1024), rng.uniform(0.05, 1, (64, 1)).astype(np.float32)], 1)
    out = {"dets": d}
    cases = {"a": ((1024, 1024), (2048, 1000), None), "b": ((1024, 1024), (683, 1024), None),
             "c": ((640, 1024), (1500, 3000), None), "d": ((1024, 1024), (4000, 4000), ((0.256, 0.256), (11.5, 3.25)))}
    for tag, (s1, s0, rp) in cases.items():
        out["out_" + tag] = scale_coords_rotated(s1, torch.from_numpy(d.copy()), s0, rp).numpy()
        out["args_" + tag] = np.array(list(s1) + list(s0) + ([rp[0][0], rp[0][1], rp[1][0], rp[1][1]] if rp else []), np.float64)
    np.savez_compressed(os.path.join(OUT, "formats.npz"), **out)
    print("formats:", {k: v.shape for k, v in out.items()})


def gen_net_forward():
    """The reference's OWN composed network run here: models.detector.S2ANet (backbone.py:283-354 -> neck.py:64-96 ->
    head.py:261-348 forward_single per level -> head.py:648-725 get_bboxes) in float32 on the CPU, batch 2 of 384 x 384
    chips (levels 48 / 24 / 12 / 6 / 3; the 48 x 48 level exceeds max_before_nms_per_level = 2000, so the top-k runs).

    Three things are patched, nothing else:
      * models.backbone.load_checkpoint (a torchvision download, backbone.py:241-255) returns the state_dict of a
        fresh ResNet; every parameter is then overwritten by tests/golden/synth_net.py anyway
      * DeformConv.forward (CUDA only, deform_conv.py:58-59 raises on the CPU) -> oracle.deform_conv_forward, so
        AlignConv.get_offset / AlignConv.forward and the ReLU stay the reference's Python
      * orn_cuda.arf_forward: the reference CPU op wraps a uint16 index at the head's [32,256,1,3,3] filter
        (SURVEY a7), so the plain definition (oracle.arf_forward) is used and checked against the reference CPU op on
        the 224-channel sub-block that stays below 65 536 elements
    The fixture holds the ordered state_dict entries (names, shapes, dtypes, checksums), the prediction-head scale
    factors that make the random network produce rotated anchors and ~1 500 NMS candidates per chip, samples of C3-C5
    and P3-P7 (every 64th / 4th element + float64 sums), the forward_single outputs of every level and get_bboxes."""
    import_reference_python()
    sys.path.insert(0, OUT)
    import synth_net
    import models.backbone as ref_backbone
    import models.dcn  # noqa: F401
    import models.orn  # noqa: F401
    ref_dc = sys.modules["models.dcn.deform_conv"]          # (models.dcn re-exports a FUNCTION of that name)
    ref_orn = ref.orn()

    ref_backbone.load_checkpoint = lambda name: ref_backbone.ResNet(name).state_dict()

    def dcn_forward(self, x, offset):
        assert self.stride == (1, 1) and self.padding == (1, 1) and self.dilation == (1, 1) and self.groups == 1
        y = oracle.deform_conv_forward(x.detach().numpy(), offset.detach().numpy(), self.weight.detach().numpy())
        return torch.from_numpy(y)
    ref_dc.DeformConv.forward = dcn_forward

    class OrnShim:
        @staticmethod
        def arf_forward(w, idx):
            full = torch.from_numpy(oracle.arf_forward(w.detach().numpy(), idx.numpy()))
            sub = w[:, :224].contiguous()
            assert sub.numel() < 65536
            assert torch.equal(ref_orn.arf_forward(sub, idx), torch.from_numpy(
                oracle.arf_forward(sub.detach().numpy(), idx.numpy()))), "plain ARF != reference CPU ARF"
            return full
        arf_backward = staticmethod(ref_orn.arf_backward)
    sys.modules["models.orn.functions.active_rotating_filter"].orn_cuda = OrnShim

    from models.detector import S2ANet
    torch.manual_seed(0)
    net = S2ANet("resnet50", 15).eval()
    sd = net.state_dict()
    names = list(sd.keys())
    shapes = [tuple(v.shape) for v in sd.values()]
    dtypes = [str(v.dtype).replace("torch.", "") for v in sd.values()]
    fixed = {n: v.numpy() for n, v in sd.items() if not v.dtype.is_floating_point and not n.endswith("num_batches_tracked")}
    B, S = 2, 384
    imgs_u8 = synth_net.synth_images(B, S, S)
    imgs = torch.from_numpy(imgs_u8).float() / 255.0                       # val.py:246-247

    def load(scales):
        st = synth_net.synth_state(names, shapes, dtypes, fixed, scales)
        net.load_state_dict(collections.OrderedDict((k, torch.from_numpy(v)) for k, v in st.items()), strict=True)
        return st

    scales = {}
    heads = {"fam_cls": ("head.fam_cls_head", 0, 1.5), "fam_reg": ("head.fam_reg_head", 1, 0.3),
             "odm_cls": ("head.odm_cls_head", 2, 1.5), "odm_reg": ("head.odm_reg_head", 3, 0.3)}
    for mod, _, _ in heads.values():
        fixed[mod + ".bias"] = np.zeros(sd[mod + ".bias"].shape, np.float32)

    def centre(p, which):
        """scale the prediction head so that its outputs spread with the wanted std around zero, per channel: the
        post-ReLU tower features have a large mean, which a random head turns into one constant per output channel"""
        mod, ki, std = heads[which]
        t = torch.cat([x.permute(1, 0, 2, 3).reshape(x.shape[1], -1) for x in p[ki]], 1).double()   # [channels, positions]
        mu = t.mean(1)
        s = std / float((t - mu[:, None]).std())
        scales[mod + ".weight"] = scales.get(mod + ".weight", 1.0) * s
        fixed[mod + ".bias"] = (-s * mu).float().numpy()

    load(scales)
    with torch.no_grad():
        p = net(imgs)["pred"]
        centre(p, "fam_reg")            # refined anchors really rotate / shift / scale ...
        centre(p, "fam_cls")
        load(scales)
        p = net(imgs)["pred"]           # ... the ODM branches see the aligned features of THOSE anchors
        centre(p, "odm_reg")
        centre(p, "odm_cls")
        load(scales)
        p = net(imgs)["pred"]
        # bias: ~1 500 (box, class) candidates per chip above 0.05, threshold placed in the widest logit gap nearby
        logits = torch.cat([t.reshape(-1) for t in p[2]]).double().numpy()
        srt = np.sort(logits)[::-1]
        kk = 1500 * B
        lo, hi = kk - 150, kk + 150
        kk = lo + int(np.argmax(srt[lo - 1:hi - 1] - srt[lo:hi]))
        q = 0.5 * (srt[kk - 1] + srt[kk])
        fixed["head.odm_cls_head.bias"] = (fixed["head.odm_cls_head.bias"] + np.float32(np.log(0.05 / 0.95) - q)).astype(np.float32)
    st = load(scales)

    feats = {}
    hooks = [net.backbone.register_forward_hook(lambda m, i, o: feats.__setitem__("C", o)),
             net.neck.register_forward_hook(lambda m, i, o: feats.__setitem__("P", o))]
    with torch.no_grad():
        pred = net(imgs)["pred"]
        res = net(imgs, post_process=True)["boxes_ls"]
    for h in hooks:
        h.remove()

    out = dict(names=np.array(names), shapes=np.array([",".join(map(str, s)) for s in shapes]), dtypes=np.array(dtypes),
               checksums=synth_net.checksums(st), seed=np.int64(synth_net.SEED), images_checksum=np.int64(imgs_u8.astype(np.int64).sum()),
               batch=np.int64(B), size=np.int64(S),
               scale_names=np.array(list(scales)), scale_values=np.array([scales[k] for k in scales], np.float64),
               fixed_names=np.array(list(fixed)))
    for k, v in fixed.items():
        out["fixed:" + k] = v
    for tag, step in (("C", 64), ("P", 4)):
        for i, t in enumerate(feats[tag]):
            a = t.numpy()
            out[f"{tag}{i}_shape"] = np.array(a.shape)
            out[f"{tag}{i}_sample"] = a.reshape(-1)[::step].copy()
            out[f"{tag}{i}_sums"] = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])
    keys = ("fam_cls", "fam_bbox", "odm_cls", "odm_bbox", "init_anchors", "refine_anchors")
    for ki, key in enumerate(keys):
        for l, t in enumerate(pred[ki]):
            out[f"{key}_{l}"] = t.numpy()
    for b, (det, lab) in enumerate(res):
        out[f"det_{b}"], out[f"labels_{b}"] = det.numpy(), lab.numpy()
        print("net_forward: image", b, "detections", det.shape[0], "labels used", len(np.unique(lab.numpy())))
    # how far the NMS decisions are from the threshold (a detection flips only if an IoU crosses it)
    np.savez_compressed(os.path.join(OUT, "net_forward.npz"), **out)
    print("net_forward: entries", len(names), "params", sum(int(np.prod(s)) for s in shapes),
          "refine angle std", float(torch.cat([t[..., 4].reshape(-1) for t in pred[5]]).std()),
          "file MB", os.path.getsize(os.path.join(OUT, "net_forward.npz")) / 1e6)


def torch_deform_conv(x, off, wgt, pad=1):
    B, C, H, W = x.shape
    O, _, kH, kW = wgt.shape
    xd, wd = x.double(), wgt.double()
    ho = torch.arange(H).view(1, H, 1).float()
    wo = torch.arange(W).view(1, 1, W).float()
    cols = []
    for i in range(kH):
        for j in range(kW):
            t = i * kW + j
            him = (ho - pad + i) + off[:, 2 * t]          # float32, as the kernel
            wim = (wo - pad + j) + off[:, 2 * t + 1]
            inside = (him > -1) & (wim > -1) & (him < H) & (wim < W)
            hl, wl = torch.floor(him), torch.floor(wim)
            lh, lw = him - hl, wim - wl
            hh, hw = 1 - lh, 1 - lw
            hl, wl = hl.long(), wl.long()
            val = torch.zeros(B, C, H, W, dtype=torch.float32)
            for (hi, wi, wt) in ((hl, wl, hh * hw), (hl, wl + 1, hh * lw), (hl + 1, wl, lh * hw), (hl + 1, wl + 1, lh * lw)):
                ok = inside & (hi >= 0) & (wi >= 0) & (hi <= H - 1) & (wi <= W - 1)
                idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(B, 1, H * W).expand(B, C, H * W)
                v = x.view(B, C, H * W).gather(2, idx).view(B, C, H, W)
                val = val + torch.where(ok.view(B, 1, H, W), wt.view(B, 1, H, W) * v, torch.zeros(()))
            cols.append(val)
    col = torch.stack(cols, 2).double()                    # [B,C,9,H,W]
    return torch.einsum("ock,bckhw->bohw", wd.view(O, C, kH * kW), col).float()


if __name__ == "__main__":
    assert build_ref.have_reference(), "run in the build container (needs /root/reference)"
    build_ref.build_all()
    if len(sys.argv) > 1:                      # e.g. `make_golden.py net_forward`: only the generators with an own rng
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    rng = np.random.default_rng(SEED)
    gen_iou(rng)
    gen_nms(rng)
    gen_arf(rng)
    gen_dcn(rng)
    gen_glue(rng)
    gen_arf_backward()
    gen_merge_nms()
    gen_dcn_backward()
    gen_assign_labels()
    gen_voc_eval()
    gen_merge_file()
    gen_rie()
    gen_net_forward()
    gen_formats()
    gen_nms_f64()
    print("done ->", OUT)
