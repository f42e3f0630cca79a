"""timing of the fused bottleneck tail (conv2 + conv3 + residual) against the two separate launches"""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2anet_amd.fused import FusedConv2d, bottleneck_tail, conv_f16

def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = "cuda"
g = torch.Generator().manual_seed(1)
B, H, W = 8, 256, 256
x = torch.relu(torch.randn(B, 64, H, W, generator=g)).to(dev).half().contiguous(memory_format=torch.channels_last)
res = torch.relu(torch.randn(B, 256, H, W, generator=g)).to(dev).half().contiguous(memory_format=torch.channels_last)
c2 = FusedConv2d(64, 64, 3, padding=1, relu=True).to(dev).half()
c3 = FusedConv2d(64, 256, 1, relu=True).to(dev).half()
w2, b2, _ = c2.packed_args(); w3, b3, _ = c3.packed_args()
x2 = torch.relu(torch.randn(B, 128, 128, 128, generator=g)).to(dev).half().contiguous(memory_format=torch.channels_last)
c4 = FusedConv2d(128, 128, 3, padding=1, relu=True).to(dev).half()
w4, b4, _ = c4.packed_args()
with torch.no_grad():
    for ph in ("1", "2"):
        os.environ["S2A_CONV_PH_NARROW"] = ph
        print(json.dumps({"ph": ph, "conv 64->64 @256^2": timeit(lambda: conv_f16(x, w2, b2, 64, 3, 1, True)),
                          "conv 128->128 @128^2": timeit(lambda: conv_f16(x2, w4, b4, 128, 3, 1, True)),
                          "fused tail": timeit(lambda: bottleneck_tail(x, c2, c3, res))}))
    del os.environ["S2A_CONV_PH_NARROW"]
    t_f = timeit(lambda: bottleneck_tail(x, c2, c3, res))
    t_a = timeit(lambda: conv_f16(x, w2, b2, 64, 3, 1, True))
    m = conv_f16(x, w2, b2, 64, 3, 1, True)
    t_b = timeit(lambda: conv_f16(m, w3, b3, 256, 1, 1, True, res))
    t_2 = timeit(lambda: conv_f16(conv_f16(x, w2, b2, 64, 3, 1, True), w3, b3, 256, 1, 1, True, res))
print(json.dumps({"fused_us": t_f, "conv2_us": t_a, "conv3_us": t_b, "two_launches_us": t_2,
                  "fused_GBps": (x.numel() + 2 * res.numel()) * 2 / t_f / 1e3}))
