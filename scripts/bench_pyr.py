"""pyramid-packed 3x3 conv / AlignConv timing (5 FPN levels of a 1024^2 chip, batch 8)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2anet_amd import pyramid as P
from s2anet_amd.pyramid import PyramidLayout
from s2anet_amd.fused import conv_pack_weight
from s2anet_amd.alignconv import pack_weight
dev = torch.device("cuda:0")
layout = PyramidLayout(8, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)], (8, 16, 32, 64, 128))
g = torch.Generator().manual_seed(0)
x = torch.randn(layout.pixels, 256, generator=g).to(dev).half()
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev).half()
b = torch.randn(256, generator=g).to(dev).half()
wp = conv_pack_weight(w)
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): f()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3
flops = 2.0 * layout.pixels * 256 * 256 * 9
us = timeit(lambda: P.conv3x3(layout, x, wp, b, 256, True))
print(json.dumps({"op": "conv3x3_pyramid", "og_env": os.environ.get("S2A_CONV_OG"), "us": round(us, 1), "tflops": round(flops / us / 1e6, 1)}))
pred = (torch.randn(layout.pixels, 64, generator=g) * 0.3).to(dev).half()
anc = P.fam_refine_anchors(layout, pred, 4.0)
wa = pack_weight(w, torch.float16)
us = timeit(lambda: P.align_conv(layout, x, anc, wa, 256))
print(json.dumps({"op": "alignconv_pyramid", "us": round(us, 1), "tflops": round(flops / us / 1e6, 1)}))
