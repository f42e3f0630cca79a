"""fused deform_conv backward (input/offset) at P3 x 8: time vs offset magnitude (how much the window slow path costs)"""
import os, sys, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scripts.bench_ops import timeit, dev
from s2anet_amd.dcn import deform_conv_backward_input_cuda
B, C, H, W, O = 8, 256, 128, 128, 256
g = torch.Generator().manual_seed(3)
x = torch.randn(B, C, H, W, generator=g).to(dev).half()
w = (torch.randn(O, C, 3, 3, generator=g) * 0.01).to(dev).half()
go = torch.randn(B, O, H, W, generator=g).to(dev).half()
gi, goff = torch.zeros_like(x), torch.zeros(B, 18, H, W, device=dev).half()
for amp in (0.0, 0.5, 1.0, 2.0, 4.0):
    off = (torch.randn(B, 18, H, W, generator=g) * amp).to(dev).half()
    t = timeit(lambda: deform_conv_backward_input_cuda(x, off, go, gi, goff, w, None, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, B), iters=5, warm=2)
    print(json.dumps(dict(offset_sigma_px=amp, ms=round(t * 1e3, 3))))
