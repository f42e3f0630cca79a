"""Round 5, item 9: how far is the float32 decode + NMS of this implementation from the reference's HALF inference path?

Under ``.half()`` (val.py:126,246) the reference decodes the boxes in c10::Half (models/head.py:684-725 ->
models/boxes.py:82-162 on half tensors) and runs ml_nms_rotated on HALF boxes and scores
(utils/ml_nms_rotated/src/nms_rotated_cuda.cu:100, AT_DISPATCH_FLOATING_TYPES_AND_HALF): a coordinate near 1 000 px sits on a
0.5 px grid, a score near 0.5 on a 2.4e-4 grid.  This implementation decodes and intersects in float32 (known
deviation).  No binary16 geometry is built here (it could not be pinned to anything: the reference has no CPU build of that
dtype); what CAN be measured is the effect of the half GRID: the bench network's candidates (batch 8, ~5 k per chip), NMS'ed
once as they are and once with the decoded boxes and the scores rounded to binary16 first (the values the reference's
half tensors would hold; the IoU arithmetic stays float32).  How many of the 8 x 2000 detections move?
-> gpurun_out/half_nms_effect.json (copied to profiles/r05_half_nms_effect.json)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from half_mode_effect import compare  # noqa: E402
from s2anet_amd import pyramid as P  # noqa: E402
from s2anet_amd.detector import build_synthetic_detector  # noqa: E402
from s2anet_amd.rotated import batched_multiclass_nms_rotated  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = build_synthetic_detector(num_classes=15, seed=1234, dtype=torch.float16, device=dev)
    g = torch.Generator(device="cpu").manual_seed(1234)
    imgs = torch.randint(0, 256, (8, 3, 1024, 1024), dtype=torch.uint8, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    got = bench.calibrate_cls_bias(model, imgs, 5000)
    head = model.head
    with torch.no_grad():
        p = model.features_to_pred(imgs, model.backbone.forward_u8(imgs, 255.0))
        layout, cls, reg, anc = p.packed
        bboxes, scores, _ = P.candidates(layout, cls, reg, anc, head.num_classes, head.max_before_nms_per_level)

    def nms(b, s):
        d, l, c = batched_multiclass_nms_rotated(b, s, head.score_thres_before_nms, head.iou_thres_nms, head.max_per_img)
        torch.cuda.synchronize()
        return d.cpu().numpy().copy(), l.cpu().numpy().copy(), c.cpu().numpy().copy()

    f32 = nms(bboxes, scores)
    f32b = nms(bboxes, scores)
    half_boxes = nms(bboxes.half().float(), scores)
    half_both = nms(bboxes.half().float(), scores.half().float())
    db = (bboxes.half().float() - bboxes).abs()
    rec = dict(workload="BASELINE configs[2]: batch 8 of 1024x1024 chips, f16 network, %.0f NMS candidates per chip" % got,
               note="decoded boxes / scores rounded to binary16 before the float32 NMS (the values the reference's half tensors "
                    "hold); the IoU arithmetic itself stays float32 -- a binary16 geometry is not built (DESIGN 2)",
               box_rounding_px=dict(max_centre=float(db[..., :2].max()), max_extent=float(db[..., 2:4].max()),
                                    max_angle_rad=float(db[..., 4].max())),
               f32_vs_half_boxes=compare(f32, half_boxes), f32_vs_half_boxes_and_scores=compare(f32, half_both),
               f32_vs_f32_noise_floor=compare(f32, f32b))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rec, open("gpurun_out/half_nms_effect.json", "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
