import sys, numpy as np, torch
sys.path.insert(0, '.')
import oracle, s2anet_amd as S
from s2anet_amd.rotated import nms_rotated_raw
rng = np.random.default_rng(1234)
def rb(n, span):
    b = np.empty((n,5), np.float32); b[:,:2]=rng.uniform(0,span,(n,2)); b[:,2:4]=rng.uniform(4,100,(n,2)); b[:,4]=rng.uniform(-np.pi/4,3*np.pi/4,n); return b
for n, span in ((8, 30), (64, 30), (65, 30), (200, 60)):
    d = rb(n, span); s = ((rng.permutation(n)+1)/(n+1)).astype(np.float32)
    D, Sc = torch.from_numpy(d).cuda(), torch.from_numpy(s).cuda()
    k = nms_rotated_raw(D, Sc, 0.3).cpu().numpy()
    ref = oracle.nms_rotated(d, s, 0.3)
    print(n, "gpu", len(k), "ref", len(ref), "equal", np.array_equal(k, ref))
    if not np.array_equal(k, ref):
        order = np.argsort(-s, kind="stable")
        iou = S.box_iou_rotated(D[order], D[order]).cpu().numpy()
        pos = {o:i for i,o in enumerate(order)}
        print(" gpu keep (sorted pos):", sorted(pos[x] for x in k)[:40])
        print(" ref keep (sorted pos):", sorted(pos[x] for x in ref)[:40])
        # emulate greedy with gpu iou matrix
        alive = np.ones(n, bool); kk=[]
        for i in range(n):
            if not alive[i]: continue
            kk.append(i); alive[(iou[i] > 0.3) & (np.arange(n) > i)] = False
        print(" emulated from gpu iou:", kk[:40])
