"""capture detect() of the synthetic detector into a HIP graph on a side stream and replay it three times (profiling target of
scripts/graph_trace.sh: which kernels a replay consists of -- there must be no fill / memset kernel and no rocPRIM partition)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2anet_amd.detector import build_synthetic_detector
dev = torch.device("cuda:0")
m = build_synthetic_detector(device=dev)
m.head.odm_cls_head.bias.data.fill_(-2.0); m.head.odm_cls_head.weight.data.mul_(20.0)
g = torch.Generator().manual_seed(5)
img = torch.randint(0, 256, (2, 3, 512, 512), dtype=torch.uint8, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.detect(img)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        out = m.detect(img)
    torch.cuda.synchronize()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
print("detections", out[2].tolist())
