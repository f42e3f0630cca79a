"""is the 3-stream step host-bound?  host enqueue time vs GPU completion time for N steps"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from s2anet_amd.detector import build_synthetic_detector
dev = torch.device("cuda", 0)
model = build_synthetic_detector(num_classes=15, seed=1234, dtype=torch.float16, device=dev)
g = torch.Generator().manual_seed(1234)
imgs = torch.randint(0, 256, (8, 3, 1024, 1024), dtype=torch.uint8, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
bench.calibrate_cls_bias(model, imgs, 5000)
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
def step(k):
    with torch.cuda.stream(streams[k % 3]):
        model.detect(imgs, max_candidates=65536)
for k in range(12): step(k)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for k in range(N): step(k)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
# single stream
for k in range(6): model.detect(imgs, max_candidates=65536)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(N): model.detect(imgs, max_candidates=65536)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("single stream: host enqueue %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
