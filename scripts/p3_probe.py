"""P3 x 8 AlignConv (1 024 tiles = four exact rounds) through both entry points and on several data sets: is the 1.6-1.7 GHz
in-kernel clock of scripts/stamps.py a property of the launch, of the data or of the measurement?"""
import os, sys, json, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2anet_amd import pyramid as P
from s2anet_amd.pyramid import PyramidLayout
from s2anet_amd.alignconv import align_conv_forward, pack_weight
dev = torch.device("cuda:0")
B, H, W, C, O, stride = 8, 128, 128, 256, 256, 8
g = torch.Generator().manual_seed(1234)
def timeit(f, warm_s=0.3, n=100):
    t = time.time()
    while time.time() - t < warm_s:
        for _ in range(20): f()
        torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): f()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3
x4 = torch.randn(B, C, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last)
ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
def anchors(jitter, sigma):
    a = torch.zeros(B, H, W, 5)
    a[..., 0] = xs * stride + 0.5 * (stride - 1) + torch.randn(B, H, W, generator=g) * jitter
    a[..., 1] = ys * stride + 0.5 * (stride - 1) + torch.randn(B, H, W, generator=g) * jitter
    a[..., 2:4] = 4 * stride * torch.exp(torch.randn(B, H, W, 2, generator=g) * sigma)
    a[..., 4] = (torch.rand(B, H, W, generator=g) - 0.25) * math.pi
    return a.to(dev)
layout = PyramidLayout(B, [(H, W)], (stride,))
xp = x4.permute(0, 2, 3, 1).reshape(-1, C).contiguous()
flops = 2.0 * O * C * 9 * B * H * W
for wscale in (0.01, 0.02):
    w = (torch.randn(O, C, 3, 3, generator=g) * wscale).to(dev).half()
    wp = pack_weight(w, torch.float16)
    for jitter, sigma in ((4.0, 0.5), (0.5, 0.1)):
        anc = anchors(jitter, sigma)
        for warm in (0.0, 0.3):
            a = timeit(lambda: align_conv_forward(x4, anc, wp, stride, relu=True, packed=True, out_channels=O), warm)
            b = timeit(lambda: P.align_conv(layout, xp, anc.reshape(-1, 5), wp, O), warm)
            print(json.dumps({"weights_sigma": wscale, "anchor_jitter": jitter, "anchor_size_sigma": sigma, "warm_s": warm,
                              "align_conv_forward_us": round(a, 1), "pyramid_entry_us": round(b, 1),
                              "mfma_frac": [round(flops / a / 1e6 / 2500, 4), round(flops / b / 1e6 / 2500, 4)]}))
