#!/usr/bin/env python3
"""GPU batched NMS vs the oracle on the candidates of the f32 end-to-end test (exact), repeated; prints what differs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
import test_gpu_e2e as T
from s2anet_amd.rotated import batched_multiclass_nms_rotated
dev = torch.device("cuda:0")
cpu, gpu, img, feats, levels, ncand = T._cpu_and_gpu_detectors(torch.float32)
with torch.no_grad():
    x = img.to(dev).float().div_(255.0)
    p = gpu.features_to_pred(x)
    bb, sc = gpu.head.candidates(p)
bbn, scn = bb[0].cpu().numpy(), sc[0].cpu().numpy()
dets_o, labels_o = oracle.multiclass_nms_rotated(bbn, scn, 0.05, 0.5, 2000)
print("oracle keeps", len(dets_o), "of", int((scn > 0.05).sum()), flush=True)
ko = np.lexsort((dets_o[:, 0], dets_o[:, 1], labels_o, -dets_o[:, 5]))
for rep in range(6):
    d, l, c = batched_multiclass_nms_rotated(bb, sc, 0.05, 0.5, 2000, None if rep % 2 == 0 else 60000)
    K = int(c[0]); gd, gl = d[0, :K].cpu().numpy(), l[0, :K].cpu().numpy()
    kg = np.lexsort((gd[:, 0], gd[:, 1], gl, -gd[:, 5]))
    same = K == len(dets_o) and np.array_equal(gd[kg].view(np.uint32), dets_o[ko].view(np.uint32)) and np.array_equal(gl[kg], labels_o[ko].astype(np.int32))
    print("rep", rep, "K", K, "identical:", same, flush=True)
    if not same:
        A = {tuple(r) for r in np.round(np.concatenate([gd, gl[:, None]], 1), 3).tolist()}
        B = {tuple(r) for r in np.round(np.concatenate([dets_o, labels_o[:, None]], 1), 3).tolist()}
        print("  only GPU:", sorted(A - B)[:6]); print("  only oracle:", sorted(B - A)[:6])
