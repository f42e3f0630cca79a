"""Round-5 item 1: where does the f16 pyramid path leave the reference network's fixture (tests/golden/net_forward.npz)?

Answers, in one run on the GPU box (everything printed as JSON lines, also written to gpurun_out/f16_fixture_diag.json):
  (a) determinism: every packed stage buffer (C3-C5, P3-P7, towers, refined anchors, AlignConv, ORConv, heads) hashed for
      two evaluations in the same process -- with the default routing (small grids -> library convolutions) and with
      S2A_OWN_CONV_ALWAYS=1 (every trunk / FPN convolution on k_conv_f16).  The hashes can be compared across boxes.
  (b) the floor-flip explanation tested instead of asserted: the f16 head evaluated three times -- with its OWN refined
      anchors, with the refined anchors of the f32 GPU network, and with the FIXTURE's refined anchors -- and the error of
      odm_cls / odm_bbox against the fixture split by "all nine sampling floors of every position in the 9 x 9 receptive
      field agree between the two anchor sets" vs the rest.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import oracle  # noqa: E402
import synth_net  # noqa: E402

OUT = []


def say(**kw):
    OUT.append(kw)
    print(json.dumps(kw), flush=True)


def fixture():
    g = np.load(os.path.join(ROOT, "tests", "golden", "net_forward.npz"), allow_pickle=False)
    names = [str(n) for n in g["names"]]
    shapes = [tuple(int(v) for v in str(s).split(",") if v) for s in g["shapes"]]
    dtypes = [str(d) for d in g["dtypes"]]
    fixed = {str(n): g["fixed:" + str(n)] for n in g["fixed_names"]}
    scales = {str(n): float(v) for n, v in zip(g["scale_names"], g["scale_values"])}
    state = synth_net.synth_state(names, shapes, dtypes, fixed, scales, seed=int(g["seed"]))
    imgs = synth_net.synth_images(int(g["batch"]), int(g["size"]), int(g["size"]), seed=int(g["seed"]))
    return dict(g=g, names=names, shapes=shapes, state=state, imgs=imgs)


def h(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def sample_floors(anc, H, W, stride):
    """floor of the nine sampling points' (h_im, w_im) of every position: [H*W, 9, 2] int (deform_conv_cuda_kernel.cu:221-228)"""
    off = oracle.align_offsets(anc.reshape(-1, 5), H, W, stride)            # [18,H,W]
    ky, kx = np.meshgrid(np.arange(3), np.arange(3), indexing="ij")
    hh = (np.arange(H)[None, :, None] - 1 + ky.reshape(-1)[:, None, None]).astype(np.float32) + off[0::2]
    ww = (np.arange(W)[None, None, :] - 1 + kx.reshape(-1)[:, None, None]).astype(np.float32) + off[1::2]
    return np.floor(hh).astype(np.int64), np.floor(ww).astype(np.int64)


def dilate(mask, r):
    out = mask.copy()
    H, W = mask.shape
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            sh = np.zeros_like(mask)
            ys, ye = max(0, dy), min(H, H + dy)
            xs, xe = max(0, dx), min(W, W + dx)
            if ye <= ys or xe <= xs:
                continue
            sh[ys:ye, xs:xe] = mask[ys - dy:ye - dy, xs - dx:xe - dx]
            out |= sh
    return out


def main():
    import test_net_forward as T
    fx = fixture()
    g = fx["g"]
    dev = torch.device("cuda:0")
    imgs = torch.from_numpy(fx["imgs"]).to(dev).contiguous(memory_format=torch.channels_last)
    strides = (8, 16, 32, 64, 128)

    def run(m, **kw):
        tr = {}
        with torch.no_grad():
            pred = m.features_to_pred(imgs, m.backbone.forward_u8(imgs, 255.0), trace=tr, **kw)
        torch.cuda.synchronize()
        return pred, tr

    def stage_hashes(tr):
        d = {f"C{i}": h(c) for i, c in enumerate(tr["C"])}
        for k in ("x", "fam_bbox", "fam_cls", "own_anchors", "align", "or_feat", "pooled", "odm_cls", "odm_bbox"):
            d[k] = h(tr[k])
        # the narrow prediction maps live in 64-column buffers: the columns the head really has
        for k, n in (("fam_bbox", 5), ("fam_cls", 15), ("odm_cls", 15), ("odm_bbox", 5)):
            d[k + "_valid"] = h(tr[k][:, :n])
            d[k + "_pad"] = h(tr[k][:, n:])
        return d

    # ---------------------------------------------------------------- (a) determinism, both routings
    m16 = T.gpu_model(fx, torch.float16)
    for mode in ("default", "own_always"):
        if mode == "own_always":
            os.environ["S2A_OWN_CONV_ALWAYS"] = "1"
        else:
            os.environ.pop("S2A_OWN_CONV_ALWAYS", None)
        import torch.nn.functional as F
        calls = {"n": 0}
        orig = F.conv2d

        def counting(*a, **k):
            calls["n"] += 1
            return orig(*a, **k)
        F.conv2d = counting
        import s2anet_amd.fused as fused
        fused.F.conv2d = counting
        try:
            _, t1 = run(m16)
        finally:
            F.conv2d = orig
            fused.F.conv2d = orig
        h1 = stage_hashes(t1)
        same_all = True
        diffs = set()
        for rep in range(3):
            _, t2 = run(m16)
            h2 = stage_hashes(t2)
            for k in h1:
                if h1[k] != h2[k]:
                    same_all = False
                    diffs.add(k)
        say(item="determinism", routing=mode, library_conv_calls=calls["n"], runs=4, bit_equal=same_all,
            stages_that_differ=sorted(diffs), hashes=h1)

    # ---------------------------------------------------------------- (b) floor flips
    os.environ["S2A_OWN_CONV_ALWAYS"] = "1"
    pred_own, tr_own = run(m16)
    layout = tr_own["layout"]
    # f32 GPU network's refined anchors (per-level path, every head op through the C ABI)
    m32 = T.gpu_model(fx, torch.float32)
    with torch.no_grad():
        p32 = m32(imgs.float().contiguous() / 255.0)["pred"]
    anc32 = torch.cat([a.reshape(-1, 5) for a in p32[4]], 0).float().contiguous()
    ancfx = torch.cat([torch.from_numpy(g[f"refine_anchors_{l}"]).reshape(-1, 5) for l in range(5)], 0).float().to(dev).contiguous()
    assert anc32.shape == tr_own["own_anchors"].shape == ancfx.shape
    del m32
    pred_32, _ = run(m16, anchors=anc32)
    pred_fx, _ = run(m16, anchors=ancfx)

    B = imgs.shape[0]
    for l in range(5):
        Hh, Ww = layout.sizes[l]
        ref_anc = g[f"refine_anchors_{l}"]
        own_anc = pred_own[4][l].float().cpu().numpy()
        flip = np.zeros((B, Hh, Ww), bool)
        for b in range(B):
            fh0, fw0 = sample_floors(ref_anc[b], Hh, Ww, strides[l])
            fh1, fw1 = sample_floors(own_anc[b], Hh, Ww, strides[l])
            flip[b] = ((fh0 != fh1) | (fw0 != fw1)).any(0)
        near = np.stack([dilate(flip[b], 4) for b in range(B)])          # 4 convolutions of 3x3 behind AlignConv
        for key, idx in (("odm_cls", 2), ("odm_bbox", 3)):
            ref = g[f"{key}_{l}"]
            rec = dict(item="floor_flip", level=l, map=key, size=[Hh, Ww], positions_with_a_flipped_floor=float(flip.mean()),
                       positions_within_reach_of_a_flip=float(near.mean()))
            for name, pr in (("own_anchors", pred_own), ("f32_gpu_anchors", pred_32), ("fixture_anchors", pred_fx)):
                err = np.abs(pr[idx][l].float().cpu().numpy() - ref)           # [B,C,H,W]
                e_clean = err[np.broadcast_to(~near[:, None], err.shape)]
                e_near = err[np.broadcast_to(near[:, None], err.shape)]
                rec[name] = dict(max=float(err.max()), mean=float(err.mean()), q99=float(np.quantile(err, 0.99)),
                                 max_clean=float(e_clean.max()) if e_clean.size else None,
                                 max_near_flip=float(e_near.max()) if e_near.size else None)
            say(**rec)
    # refined anchors of the f16 path vs fixture, in units of the box
    for l in range(5):
        Hh, Ww = layout.sizes[l]
        ref = g[f"refine_anchors_{l}"]
        got = pred_own[4][l].float().cpu().numpy()
        d = np.abs(got - ref)
        d[..., 4] = np.minimum(d[..., 4], np.abs(d[..., 4] - np.pi))
        size = np.maximum(1, np.maximum(ref[..., 2], ref[..., 3]))[..., None]
        fb = np.abs(pred_own[1][l].float().cpu().numpy() - g[f"fam_bbox_{l}"])
        fc = np.abs(pred_own[0][l].float().cpu().numpy() - g[f"fam_cls_{l}"])
        # how far a sampling point moves, in feature pixels: centre shift + extent change + rotation of the outer taps
        reach = np.hypot(ref[..., 2], ref[..., 3]) / 2 / strides[l]
        move = (np.hypot(d[..., 0], d[..., 1]) + np.hypot(d[..., 2], d[..., 3]) / 2) / strides[l] + d[..., 4] * reach
        say(item="anchors", level=l, centre_extent_err_over_box_max=float((d[..., :4] / size).max()),
            q99=float(np.quantile(d[..., :4] / size, 0.99)), angle_max=float(d[..., 4].max()),
            centre_px_max=float(d[..., :2].max() / strides[l]), sample_move_px_max=float(move.max()),
            sample_move_px_mean=float(move.mean()), fam_bbox_err_max=float(fb.max()), fam_bbox_err_mean=float(fb.mean()),
            fam_bbox_abs_mean=float(np.abs(g[f"fam_bbox_{l}"]).mean()), fam_cls_err_max=float(fc.max()),
            P_abs_mean=float(tr_own["x"][layout.pix0[l]:layout.pix0[l] + B * Hh * Ww].float().abs().mean()))
    # detections of the three variants against the fixture
    for name, kw in (("own_anchors", {}), ("fixture_anchors", dict(anchors=ancfx))):
        with torch.no_grad():
            p = m16.features_to_pred(imgs, m16.backbone.forward_u8(imgs, 255.0), **kw)
            dets, labels, counts = m16.head.get_bboxes_batched(p)
        for b in range(B):
            rd, rl = g[f"det_{b}"], g[f"labels_{b}"]
            k = int(counts[b])
            d, lab = dets[b, :k].cpu().numpy().astype(np.float64), labels[b, :k].cpu().numpy()
            hit = 0
            for j in range(len(rd)):
                c = (lab == rl[j]) & (np.abs(d[:, 5] - rd[j, 5]) < 2e-2) & (np.abs(d[:, :2] - rd[j, :2]).max(1) < 1.0)
                hit += bool(c.any())
            say(item="detections", anchors=name, image=b, reference=len(rd), got=k, matched=hit)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "f16_fixture_diag.json"), "w") as f:
        for r in OUT:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
