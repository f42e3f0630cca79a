"""detect() on a range of image sizes / batch sizes, channels-last and NCHW uint8 input (no crashes, finite boxes)"""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2anet_amd.detector import build_synthetic_detector
dev = torch.device("cuda:0")
m = build_synthetic_detector(device=dev)
m.head.odm_cls_head.bias.data.fill_(-1.5); m.head.odm_cls_head.weight.data.mul_(10)
for (B, H, W) in ((2, 1024, 768), (3, 640, 640), (1, 512, 1280), (2, 800, 800), (1, 416, 608), (5, 1024, 1024), (1, 384, 384)):
    img = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device=dev).contiguous(memory_format=torch.channels_last)
    d, l, c = m.detect(img)
    torch.cuda.synchronize()
    d2, l2, c2 = m.detect(img.contiguous())      # NCHW uint8: stock stem path
    torch.cuda.synchronize()
    print((B, H, W), "counts", c.tolist(), "nchw counts", c2.tolist(), "finite", bool(torch.isfinite(d).all()))
