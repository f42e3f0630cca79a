import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import s2anet_amd as S, oracle
rng = np.random.default_rng(3)
B, n, C = 8, 5344, 15
boxes = np.stack([np.concatenate([rng.uniform(0, 1024, (n, 2)), rng.uniform(8, 80, (n, 2)), rng.uniform(-0.7, 2.3, (n, 1))], 1) for _ in range(B)]).astype(np.float32)
scores = (rng.random((B, n, C)) ** 12).astype(np.float32)
bb, sc = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
for cap in (None, 160000):
    for _ in range(3): out = S.batched_multiclass_nms_rotated(bb, sc, 0.05, 0.5, 2000, max_candidates=cap)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): out = S.batched_multiclass_nms_rotated(bb, sc, 0.05, 0.5, 2000, max_candidates=cap)
    torch.cuda.synchronize(); print("cap", cap, "ms per call", (time.perf_counter() - t) / 10 * 1e3, "cand", int((scores > 0.05).sum()))
d, l, c = out
for b in (0, 7):
    rd, rl = oracle.multiclass_nms_rotated(boxes[b], scores[b], 0.05, 0.5, 2000)
    k = int(c[b]); assert k == len(rd) and np.array_equal(d[b, :k].cpu().numpy(), rd), b
print("ok")
