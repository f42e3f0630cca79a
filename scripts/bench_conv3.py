"""3x3 layers on small maps: 8x16 vs 4x16 tiles (S2A_CONV3_HALF)"""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2anet_amd.fused import conv_f16, conv_pack_weight
def timeit(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
g = torch.Generator().manual_seed(1)
for (B, C, H, W, O) in ((8, 256, 64, 64, 256), (8, 512, 32, 32, 512), (8, 256, 32, 32, 256), (8, 128, 128, 128, 128), (8, 256, 128, 128, 256)):
    x = torch.relu(torch.randn(B, C, H, W, generator=g)).to("cuda").half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(O, C, 3, 3, generator=g) * 0.03).to("cuda").half()
    b = torch.randn(O, generator=g).to("cuda").half()
    wp = conv_pack_weight(w)
    out = torch.empty((B, O, H, W), dtype=torch.float16, device="cuda", memory_format=torch.channels_last)
    res = {}
    for half in ("0", "1"):
        os.environ["S2A_CONV3_HALF"] = half
        res[half] = round(timeit(lambda: conv_f16(x, wp, b, O, 3, 1, True, out=out)), 1)
    print(json.dumps({"shape": [B, C, H, W, O], "us_8x16": res["0"], "us_4x16": res["1"], "TF_best": round(2.0 * B * H * W * C * O * 9 / min(res.values()) / 1e6, 1)}))
