"""ml_nms_rotated on a detector-like input: 160 000 rows of which 40 000 are real (the rest ignored padding does not exist in
the drop-in op, so: 40 000 rows x 120 labels, clustered so that tiles are dense), forced through the big-input cull"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["S2A_NMS_CULL_LANES"] = "1"
os.environ["S2A_NMS_SPATIAL"] = "0"
import numpy as np, torch
from s2anet_amd.rotated import ml_nms_rotated
rng = np.random.default_rng(7)
n, nl = 40000, 120
lab = rng.integers(0, nl, n)
cen = rng.uniform(100, 900, (nl, 6, 2))                        # 6 clusters per label
k = rng.integers(0, 6, n)
d = np.empty((n, 5), np.float32)
d[:, :2] = cen[lab, k] + rng.normal(0, 12, (n, 2))
d[:, 2:4] = rng.uniform(20, 60, (n, 2)); d[:, 4] = rng.uniform(-0.7, 2.3, n)
s = ((rng.permutation(n) + 1) / (n + 1)).astype(np.float32)
D, S_, L_ = torch.from_numpy(d).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(lab.astype(np.float32)).cuda()
for _ in range(3): kk = ml_nms_rotated(D, S_, L_, 0.5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ml_nms_rotated(D, S_, L_, 0.5)
e1.record(); torch.cuda.synchronize()
print(json.dumps(dict(op="ml_nms dense detector-like", n=n, labels=nl, keep=int(kk.numel()), ms=round(e0.elapsed_time(e1) / 10, 4))))
