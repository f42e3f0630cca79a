"""the f32 GPU network against the reference-network fixture, several evaluations with and without
torch.backends.cudnn.deterministic: which bound does the composed f32 head hold, and is the spread run-to-run?
python scripts/f32_fixture_probe.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import test_net_forward as T
from conftest import golden
g = golden("net_forward.npz")
fxd = T.fx._get_wrapped_function()()
imgs = torch.from_numpy(fxd["imgs"]).cuda()
names = ("fam_cls", "fam_bbox", "odm_cls", "odm_bbox", "refine_anchors")
for det in (False, True, False, True):
    torch.backends.cudnn.deterministic = det
    torch.backends.cudnn.benchmark = False
    m = T.gpu_model(fxd, torch.float32)
    with torch.no_grad():
        pred = m(imgs.float() / 255.0)["pred"]
    out = {}
    for key, per_level in zip(names, pred):
        e = 0.0
        for l, t in enumerate(per_level):
            a, b = t.cpu().numpy().astype(np.float64), g[f"{key}_{l}"].astype(np.float64)
            e = max(e, float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max()))
        out[key] = e
    print("deterministic=%s  " % det + "  ".join("%s %.2e" % kv for kv in out.items()), flush=True)
