"""End-to-end effect of the reference-half sampling mode (S2A_DCN_HALF_COORDS=1) on the bench network: BASELINE
configs[2] (batch 8 of 1024 x 1024 chips, f16, calibrated classifier, ~5 k NMS candidates per chip), the same batch
through detect() with the AlignConv sampling coordinates in f32 (default) and rounded as the reference's Half
instantiation rounds them.  How many of the 8 x 2000 detections differ?  -> gpurun_out/half_mode_effect.json
(copied to profiles/r04_half_mode_effect.json).  A detection "matches" when the other run holds one of the same label
whose centre is within 1 px and whose score is within 0.02; "identical" = bit-equal row.  The trunk's two library
convolutions are not run-to-run deterministic, so the same comparison between two DEFAULT runs is the noise floor."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from s2anet_amd.detector import build_synthetic_detector  # noqa: E402


def compare(a, b):
    (da, la, ca), (db, lb, cb) = a, b
    out = dict(images=int(da.shape[0]), detections_a=int(ca.sum()), detections_b=int(cb.sum()), identical=0, matched=0,
               unmatched_a=0, unmatched_b=0, max_score_shift=0.0, mean_centre_shift_px=0.0)
    shifts = []
    for i in range(da.shape[0]):
        A, B = da[i, :ca[i]], db[i, :cb[i]]
        LA, LB = la[i, :ca[i]], lb[i, :cb[i]]
        setb = {(int(l),) + tuple(r.view(np.uint32).tolist()) for r, l in zip(B, LB)}
        out["identical"] += sum(((int(l),) + tuple(r.view(np.uint32).tolist())) in setb for r, l in zip(A, LA))
        used = np.zeros(len(B), bool)
        for r, l in zip(A, LA):
            c = np.nonzero((LB == l) & ~used & (np.abs(B[:, 5] - r[5]) < 0.02) &
                           (np.abs(B[:, :2] - r[:2]).max(1) < 1.0))[0]
            if len(c):
                j = c[np.argmin(np.abs(B[c, :2] - r[:2]).sum(1))]
                used[j] = True
                out["matched"] += 1
                out["max_score_shift"] = max(out["max_score_shift"], float(abs(B[j, 5] - r[5])))
                shifts.append(float(np.hypot(*(B[j, :2] - r[:2]))))
            else:
                out["unmatched_a"] += 1
        out["unmatched_b"] += int((~used).sum())
    out["mean_centre_shift_px"] = float(np.mean(shifts)) if shifts else 0.0
    return out


def main():
    dev = torch.device("cuda", 0)
    model = build_synthetic_detector(num_classes=15, seed=1234, dtype=torch.float16, device=dev)
    g = torch.Generator(device="cpu").manual_seed(1234)
    imgs = torch.randint(0, 256, (8, 3, 1024, 1024), dtype=torch.uint8, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    got = bench.calibrate_cls_bias(model, imgs, 5000)

    def run(half):
        if half:
            os.environ["S2A_DCN_HALF_COORDS"] = "1"
        else:
            os.environ.pop("S2A_DCN_HALF_COORDS", None)
        with torch.no_grad():
            d, l, c = model.detect(imgs)
        torch.cuda.synchronize()
        os.environ.pop("S2A_DCN_HALF_COORDS", None)
        return d.cpu().numpy().copy(), l.cpu().numpy().copy(), c.cpu().numpy().copy()

    base, base2, half = run(False), run(False), run(True)
    rec = dict(workload="BASELINE configs[2]: batch 8 of 1024x1024 chips, f16, %.0f NMS candidates per chip" % got,
               default_vs_half_coords=compare(base, half), default_vs_default_noise_floor=compare(base, base2))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rec, open("gpurun_out/half_mode_effect.json", "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
