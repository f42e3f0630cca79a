"""one process, 8 ml_nms_rotated calls at 200 k rows x 15 labels (profiling target of scripts/pmc_cmd.sh / nms_timeline.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scripts.bench_ops import rboxes
from s2anet_amd.rotated import ml_nms_rotated
rng = np.random.default_rng(1234)
n = 200000
d = torch.from_numpy(rboxes(rng, n)).cuda(); s = torch.from_numpy(rng.permutation(n).astype(np.float32) / n).cuda()
l = torch.from_numpy(rng.integers(0, 15, n).astype(np.float32)).cuda()
for _ in range(8): ml_nms_rotated(d, s, l, 0.5)
torch.cuda.synchronize()
