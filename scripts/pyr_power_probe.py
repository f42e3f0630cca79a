"""Is the pyramid AlignConv launch paced by the clock the chip holds (DVFS) rather than by its instruction stream?  Same
launch, same instruction stream, three data sets: all zeros, ReLU-sparse (what the detector's step feeds it), dense random.
Equal cycles but different times = clock-limited (MI355X_MICROARCH.md, DVFS give-back (1))."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2anet_amd import pyramid as P
from s2anet_amd.pyramid import PyramidLayout
from s2anet_amd.alignconv import pack_weight
dev = torch.device("cuda:0")
layout = PyramidLayout(8, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)], (8, 16, 32, 64, 128))
g = torch.Generator().manual_seed(0)
xr = torch.randn(layout.pixels, 256, generator=g).to(dev).half()
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev).half()
pred = (torch.randn(layout.pixels, 64, generator=g) * 0.3).to(dev).half()
anc = P.fam_refine_anchors(layout, pred, 4.0)
def timeit(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): f()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3
for name, x, ww in (("zeros", torch.zeros_like(xr), torch.zeros_like(w)), ("relu-sparse", torch.relu(xr), w), ("dense", xr, w)):
    wa = pack_weight(ww, torch.float16)
    us = timeit(lambda: P.align_conv(layout, x, anc, wa, 256))
    print(json.dumps({"kernel": "k_dcn_patch", "data": name, "us": round(us, 1),
                      "mfma_frac": round(2.0 * layout.pixels * 256 * 2304 / us / 1e6 / 2500, 4)}))
