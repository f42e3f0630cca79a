"""Same-box A/B of the head's pyramid-packed 3x3 256 -> 256 launch: direct kernel (k_conv_f16<9,4,2>) against the Winograd
F(2,3)-along-x kernel (k_conv_wino_f16), alternating, on zeros / ReLU-sparse / dense data (the clock the chip holds depends on
the data: MI355X_MICROARCH.md, DVFS give-back).  One JSON line per (kernel, data)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2anet_amd import pyramid as P
from s2anet_amd.pyramid import PyramidLayout
from s2anet_amd.fused import conv_pack_weight, conv_wino_pack_weight
dev = torch.device("cuda:0")
layout = PyramidLayout(8, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)], (8, 16, 32, 64, 128))
g = torch.Generator().manual_seed(0)
xr = torch.randn(layout.pixels, 256, generator=g).to(dev).half()
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev).half()
b = torch.randn(256, generator=g).to(dev).half()
out = layout.new(256, dev)


def timeit(f, n=200):
    for _ in range(30): f()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): f()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3


flop_direct = 2.0 * layout.pixels * 256 * 2304
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for name, x, ww in (("zeros", torch.zeros_like(xr), torch.zeros_like(w)), ("relu-sparse", torch.relu(xr), w), ("dense", xr, w)):
    wd, wu = conv_pack_weight(ww), conv_wino_pack_weight(ww)
    res = {"direct": [], "wino": []}
    for _ in range(rounds):
        res["direct"].append(timeit(lambda: P.conv3x3(layout, x, wd, b, 256, relu=True, out=out)))
        res["wino"].append(timeit(lambda: P.conv3x3_wino(layout, x, wu, b, 256, relu=True, out=out)))
    for k, v in res.items():
        us = min(v)
        issued = flop_direct * (6.0 / 9.0 if k == "wino" else 1.0)
        print(json.dumps({"kernel": k, "data": name, "us": round(us, 1), "all_us": [round(t, 1) for t in v],
                          "direct_equivalent_TFLOPs": round(flop_direct / us / 1e6, 1),
                          "mfma_frac_issued": round(issued / us / 1e6 / 2500, 4)}), flush=True)
