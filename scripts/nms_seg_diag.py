#!/usr/bin/env python3
"""segmented NMS vs oracle under the debug switches: python scripts/nms_seg_diag.py n nseg span"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
from conftest import rand_rboxes, distinct_scores
from s2anet_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
n, nseg, span = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
rng = np.random.default_rng(1234)
d = rand_rboxes(rng, n, span=span); s = distinct_scores(rng, n); seg = rng.integers(0, nseg, n).astype(np.int32)
ref = np.zeros(n, bool)
for c in range(nseg):
    idx = np.nonzero(seg == c)[0]
    ref[idx[oracle.nms_rotated(d[idx], s[idx], 0.5)]] = True
D, S, G = [torch.from_numpy(a).to(dev) for a in (d, s, seg)]
ws = torch.empty(L.s2a_nms_rotated_workspace_bytes(n, n), dtype=torch.uint8, device=dev)
for sp in ("0", "1"):
    for gl in ("0", "1"):
        os.environ["S2A_NMS_SPATIAL"] = sp; os.environ["S2A_NMS_FINISH_GLOBAL"] = gl
        flags = torch.zeros(n, dtype=torch.uint8, device=dev)
        _lib.check(L.s2a_nms_rotated_segmented(_lib.ptr(D), _lib.ptr(S), _lib.ptr(G), None, n, nseg, 1, 0.5, _lib.ptr(flags), None, None, 0,
                                               _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
        got = flags.cpu().numpy().astype(bool)
        print("n %d nseg %d spatial %s finish_global %s: mismatches %d (kept %d, ref %d)" % (n, nseg, sp, gl, int((got != ref).sum()), got.sum(), ref.sum()), flush=True)
