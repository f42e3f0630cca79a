#!/usr/bin/env python3
"""prints the stage-by-stage differences the end-to-end parity tests assert on (tests/test_gpu_e2e.py), for both
dtypes, so that their tolerances are set from measurements: python scripts/e2e_diag.py [f32|f16|both]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
from oracle import pipeline
import test_gpu_e2e as T

which = sys.argv[1] if len(sys.argv) > 1 else "both"
dev = torch.device("cuda:0")

def stats(name, a, b):
    e = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    print("  %-28s max %.3e mean %.3e (ref absmax %.3e, std %.3e)" % (name, e.max(), e.mean(), np.abs(b).max(), np.std(b)), flush=True)

if which in ("f32", "both"):
    t0 = time.time()
    cpu, gpu, img, feats, levels, ncand = T._cpu_and_gpu_detectors(torch.float32)
    print("f32: cpu pipeline + build %.1f s, candidates %d" % (time.time() - t0, ncand), flush=True)
    with torch.no_grad():
        x = img.to(dev).float().div_(255.0)
        c3 = gpu.backbone(x)
        gf = gpu.neck(c3)
        for li, (g, c) in enumerate(zip(gf, feats)):
            stats("fpn level %d" % li, g.cpu().numpy(), c.numpy())
        p = gpu.features_to_pred(x)
        for li, lv in enumerate(levels):
            H, W = lv["size"]
            stats("refined anchors L%d" % li, p[4][li][0].reshape(-1, 5).cpu().numpy(), lv["refined"])
            stats("cls logits L%d" % li, p[2][li][0].permute(1, 2, 0).reshape(-1, 15).cpu().numpy(), lv["cls"])
            stats("reg deltas L%d" % li, p[3][li][0].permute(1, 2, 0).reshape(-1, 5).cpu().numpy(), lv["reg"])
        d, l, c, ovf = gpu.detect(img.to(dev), return_overflow=True)
    dets_c, labels_c, bb_c, sc_c = pipeline.postprocess(levels)
    K = int(c[0])
    print("  GPU: %d detections, candidates %s ; CPU: %d detections, candidates %d" % (K, ovf.cpu().tolist(), len(dets_c), int((sc_c > 0.05).sum())))
    gd, gl = d[0, :K].cpu().numpy(), l[0, :K].cpu().numpy()
    pairs, lg, lc = T._match(gd, gl, dets_c, labels_c.astype(np.int32))
    print("  matched %d, unmatched GPU %d / CPU %d, order swaps %d" % (len(pairs), len(lg), len(lc), sum(1 for i, j in pairs if i != j)))
    # margins on the CPU side: how close decisions are to their thresholds
    sc = sc_c.reshape(-1)
    print("  min |score - 0.05| = %.3e ; scores within 1e-5 of thr: %d" % (np.abs(sc - 0.05).min(), int((np.abs(sc - 0.05) < 1e-5).sum())))
    del gpu
    torch.cuda.empty_cache()

if which in ("f16", "both"):
    from s2anet_amd import pyramid as P
    t0 = time.time()
    cpu, gpu, img, feats, levels, ncand = T._cpu_and_gpu_detectors(torch.float16)
    print("f16: cpu pipeline + build %.1f s, candidates %d" % (time.time() - t0, ncand), flush=True)
    imgs = img.to(dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        c3 = gpu.backbone.forward_u8(imgs, 255.0)
        from s2anet_amd.pyramid import PyramidLayout
        p = gpu.features_to_pred(imgs, c3)
    layout, cls, reg, anc = p.packed
    for li, lv in enumerate(levels):
        H, W = lv["size"]
        stats("refined anchors L%d" % li, layout.rows(anc, li).view(H * W, 5).cpu().numpy(), lv["refined"])
        stats("cls logits L%d" % li, layout.level(cls, li, 15)[0].permute(1, 2, 0).reshape(-1, 15).float().cpu().numpy(), lv["cls"])
        stats("reg deltas L%d" % li, layout.level(reg, li, 5)[0].permute(1, 2, 0).reshape(-1, 5).float().cpu().numpy(), lv["reg"])
    with torch.no_grad():
        d, l, c, ovf = gpu.head.get_bboxes_batched(p, return_overflow=True)
    dets_c, labels_c, _, sc_c = pipeline.postprocess(levels, half_scores=True)
    print("  GPU: %d detections, candidates %s ; CPU(f32 math on f16 params): %d detections, candidates %d"
          % (int(c[0]), ovf.cpu().tolist(), len(dets_c), int((sc_c > 0.05).sum())))
