"""1x1 layers of the trunk's deep stages: 128- vs 64-position tiles (S2A_CONV1_HALF)"""
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
shapes = [(8, 1024, 64, 64, 256, 1, False), (8, 256, 64, 64, 1024, 1, True), (8, 2048, 32, 32, 512, 1, False),
          (8, 512, 32, 32, 2048, 1, True), (8, 1024, 64, 64, 512, 1, False), (8, 1024, 64, 64, 2048, 2, False),
          (8, 2048, 32, 32, 256, 1, False), (8, 512, 128, 128, 256, 1, False)]
for (B, C, H, W, O, st, with_res) in shapes:
    x = torch.relu(torch.randn(B, C, H, W, generator=g)).to("cuda").half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(O, C, 1, 1, generator=g) * 0.03).to("cuda").half()
    b = torch.randn(O, generator=g).to("cuda").half()
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    r = torch.randn(B, O, Ho, Wo, generator=g).to("cuda").half().contiguous(memory_format=torch.channels_last) if with_res else None
    wp = conv_pack_weight(w)
    out = torch.empty((B, O, Ho, Wo), dtype=torch.float16, device="cuda", memory_format=torch.channels_last)
    res = {}
    for half in ("0", "1"):
        os.environ["S2A_CONV1_HALF"] = half
        res[half] = round(timeit(lambda: conv_f16(x, wp, b, O, 1, st, True, r, out=out)), 1)
    os.environ["S2A_CONV1_HALF"] = "0"
    for og in ("2", "1"):          # measurement builds only (-DS2A_MEASURE): narrower out-channel groups
        os.environ["S2A_CONV1_OG"] = og
        res["og" + og] = round(timeit(lambda: conv_f16(x, wp, b, O, 1, st, True, r, out=out)), 1)
    os.environ.pop("S2A_CONV1_OG")
    flops = 2.0 * B * Ho * Wo * C * O
    bytes_ = 2.0 * B * (H * W * C + Ho * Wo * O * (2 if with_res else 1)) + 2.0 * C * O
    print(json.dumps({"shape": [B, C, H, W, O, st], "res": with_res, "us_128": res["0"], "us_64": res["1"], "us_og2": res["og2"],
                      "us_og1": res["og1"], "floor_us": round(max(flops / 2.5e9, bytes_ / 8e6), 1),
                      "TF_best": round(flops / min(res.values()) / 1e6, 1)}))
