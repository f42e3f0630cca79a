import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle, s2anet_amd as S
from conftest import rand_rboxes
rng = np.random.default_rng(1234)
B, n, C = 3, 700, 15
boxes = np.stack([rand_rboxes(rng, n, span=260) for _ in range(B)])
scores = (rng.random((B, n, C)) ** 8).astype(np.float32)
scores[2] *= 0.01
for cap in (None, 4000):
    dets, labels, counts = S.batched_multiclass_nms_rotated(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.05, 0.5, 200, max_candidates=cap)
    print("cap", cap, "counts", counts.tolist())
    for b in range(B):
        rd, rl = oracle.multiclass_nms_rotated(boxes[b], scores[b], 0.05, 0.5, 200)
        kb = int(counts[b]); d = dets[b,:kb].cpu().numpy()
        print(" img", b, "ref", rd.shape[0], "gpu", kb, "ncand", (scores[b]>0.05).sum())
        if kb == rd.shape[0] and kb:
            bad = np.nonzero((d != rd).any(1))[0]
            print("  mismatching rows:", len(bad), bad[:10])
            if len(bad):
                i = bad[0]; print("  gpu", d[i], labels[b,i].item(), "\n  ref", rd[i], rl[i])
                print("  score sorted desc gpu?", (np.diff(d[:,5])<=0).all(), " ref?", (np.diff(rd[:,5])<=0).all())
                print("  same set?", set(map(tuple, d.round(4))) == set(map(tuple, rd.round(4))))
