#!/usr/bin/env python3
"""step-by-step hipGraph capture / replay of the head + post-processing, printing after every step.
python scripts/graph_diag.py <variant>: a = torch.cuda.graph default stream, no cap; b = default stream, cap 50000;
c = explicit side stream, no cap; d = candidates + multiclass_candidates only (no NMS), side stream, cap 50000;
e = like the failing case but cap = 200000 (> candidates)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2anet_amd import _lib
from s2anet_amd.detector import build_synthetic_detector
from s2anet_amd.pyramid import PyramidLayout
def say(*a): print(*a, flush=True)
dev = torch.device("cuda:0")
v = sys.argv[1]
m = build_synthetic_detector(device=dev)
m.head.odm_cls_head.bias.data.fill_(-2.0); m.head.odm_cls_head.weight.data.mul_(20.0)
layout = PyramidLayout(2, [(40, 48), (20, 24), (10, 12), (5, 6), (3, 3)], (8, 16, 32, 64, 128))
g = torch.Generator().manual_seed(11)
feats = [torch.randn(layout.pixels, 256, generator=g).to(dev).half() for _ in range(3)]
cap = {"a": None, "b": 50000, "c": None, "d": 50000, "e": 200000, "d2": 50000, "d3": 50000, "d4": 50000}[v]
def run(x):
    p = m.head.forward_pyramid(layout, x)
    if v[0] == "d":
        from s2anet_amd import pyramid as P
        import ctypes
        lay, cls, reg, anc = p.packed
        if v == "d2":            # select only, on persistent inputs made outside the capture
            bb, sc = PERSIST
        else:
            bb, sc, sel = P.candidates(lay, cls, reg, anc, 15, 2000)
            if v == "d3":        # candidates only
                return (bb, sc, sel.float())
            if v == "d4":        # candidates, then select on persistent copies of their outputs
                PERSIST[0].copy_(bb); PERSIST[1].copy_(sc)
                bb, sc = PERSIST
        B, n, C = sc.shape
        total = B * n * C
        L = _lib.lib()
        c = min(cap, total)
        outs = [torch.empty((c, 5), dtype=torch.float32, device=dev), torch.empty((c,), dtype=torch.float32, device=dev)] + \
               [torch.empty((c,), dtype=torch.int32, device=dev) for _ in range(3)] + [torch.empty((1,), dtype=torch.int64, device=dev)]
        ws = _lib.workspace(L.s2a_multiclass_candidates_workspace_bytes(total), dev, "cand")
        _lib.check(L.s2a_multiclass_candidates(_lib.ptr(bb.reshape(-1, 5)), _lib.ptr(sc.reshape(-1)), B, n, C, 0.05, c,
                                               *[_lib.ptr(o) for o in outs], _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
        return tuple(outs) + (bb, sc)
    return m.head.get_bboxes_batched(p, max_candidates=cap, return_overflow=True)
PERSIST = [torch.rand(2, 2559, 5, device=dev) * 100 + 10, torch.rand(2, 2559, 15, device=dev)]
with torch.no_grad():
    eager = [tuple(t.clone() for t in run(x)) for x in feats]
    torch.cuda.synchronize(); say(v, "eager ok; candidates", [(int(e[3][0]) if v[0] != "d" else (int(e[5][0]) if v != "d3" else -1)) for e in eager])
    static_x = feats[0].clone()
    graph = torch.cuda.CUDAGraph()
    if v in ("a", "b"):
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run(static_x)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            out = run(static_x)
    else:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run(static_x)
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            out = run(static_x)
    torch.cuda.synchronize(); say("captured")
    for k in (0, 0, 1, 2, 0):
        static_x.copy_(feats[k])
        graph.replay()
        torch.cuda.synchronize()
        say("replay", k, "equal:", all(torch.equal(a, b) for a, b in zip(out[:3], eager[k][:3])))
say("DONE", v)
