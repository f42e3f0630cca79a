"""one process, 30 ml_nms_rotated calls at n rows x 15 labels (default 5000): profiling target for the small-input path"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scripts.bench_ops import rboxes
from s2anet_amd.rotated import ml_nms_rotated
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = np.random.default_rng(1234)
d = torch.from_numpy(rboxes(rng, n)).cuda(); s = torch.from_numpy(((rng.permutation(n) + 1) / (n + 1)).astype(np.float32)).cuda()
l = torch.from_numpy(rng.integers(0, 15, n).astype(np.float32)).cuda()
for _ in range(10): ml_nms_rotated(d, s, l, 0.5)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): ml_nms_rotated(d, s, l, 0.5)
torch.cuda.synchronize()
print("host wall per call: %.1f us" % ((time.perf_counter() - t0) / 30 * 1e6))
if os.environ.get("S2A_ALLOW_MEASURE_BUILD"):
    import ctypes
    from s2anet_amd import _lib
    L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "s2anet_amd", "libs2anet_hip.so"))
    buf = (ctypes.c_ulonglong * (64 * 16))()
    L.s2a_debug_small_stamps(buf)
    st = np.array(buf, dtype=np.int64).reshape(64, 16)
    names = ["start", "labels done", "rows+keys", "sorted", "boxes", "cull", "exact", "greedy", "merge loaded", "merge sorted"]
    for b in range(16):
        r = st[b]
        if r[0] == 0: continue
        print("wg %2d " % b + " | ".join("%s %d" % (names[k], r[k] - r[k - 1]) for k in range(1, 10) if r[k] > 0 and r[k - 1] > 0))
