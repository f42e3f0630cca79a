"""GPU parity at the sizes BASELINE.json names (VERDICT r01 "parity-size gaps"):
  * configs[1]: AlignConv forward on one 256 x 128 x 128 FPN level with SURVEY 8(d) config-2 anchors, f32 <= 1e-4
    against the oracle on the whole image, and the benchmarked f16 instantiation (k_dcn_patch, C = O = 256)
  * configs[2]: the whole detector, image -> detections, against the CPU pipeline (oracle/pipeline.py) on one
    1024 x 1024 chip, plus the exact post-processing chain from the GPU's own intermediate tensors
  * hipGraph capture of detect() replayed on changed inputs == eager
  * the candidate cap of the batched NMS reports what it dropped
Everything product-side goes through the C ABI (s2anet_amd/_lib.py)."""
import math

import numpy as np
import pytest
import torch

import oracle
from oracle import pipeline

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def config2_inputs(B, C=256, H=128, W=128, O=256, stride=8, seed=1234):
    """x ~ N(0,1); anchors = stride-8 grid anchors perturbed by dxy ~ N(0, 4 px),
    w,h = 32 exp(N(0, 0.5)), theta ~ U(-pi/4, 3pi/4); weight ~ N(0, 0.01) (alignconv.py:25-26)"""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    anc = np.stack([oracle.grid_anchors(H, W, stride) for _ in range(B)]).reshape(B, H, W, 5).copy()
    anc[..., 0:2] += rng.normal(0, 4, anc[..., 0:2].shape)
    anc[..., 2:4] = 32 * np.exp(rng.normal(0, 0.5, anc[..., 2:4].shape))
    anc[..., 4] = rng.uniform(-np.pi / 4, 3 * np.pi / 4, anc[..., 4].shape)
    w = (rng.standard_normal((O, C, 3, 3)) * 0.01).astype(np.float32)
    return x, anc.astype(np.float32), w


def test_config1_alignconv_f32_full_size_vs_oracle():
    """BASELINE configs[1] as stated: one P3 level [1,256,128,128], f32, whole image <= 1e-4 (north_star tolerance)"""
    import s2anet_amd as S
    x, anc, w = config2_inputs(1)
    ac = S.AlignConv(256, 256, 3).to(dev())
    with torch.no_grad():
        ac.deform_conv.weight.copy_(cu(w))
        out = ac(cu(x), cu(anc), 8).cpu().numpy()                          # NCHW f32 in, the reference layout
        out_cl = ac(cu(x).contiguous(memory_format=torch.channels_last), cu(anc), 8).cpu().numpy()
    off = oracle.align_offsets(anc[0].reshape(-1, 5), 128, 128, 8)[None]
    ref = oracle.deform_conv_forward(x, off, w, relu=True)
    err = np.abs(out - ref)
    assert out.shape == ref.shape == (1, 256, 128, 128)
    assert err.max() < 1e-4, err.max()
    assert np.abs(out_cl - ref).max() < 1e-4
    assert (ref > 0).mean() > 0.3 and np.abs(ref).max() > 1.0              # a real signal, not zeros


def test_config1_alignconv_f16_production_kernel_vs_oracle():
    """the instantiation the benchmark runs -- k_dcn_patch<NHWC, anchors>, C = O = 256, f16, batch > 1 -- on config-2
    inputs against the f16-column oracle on the WHOLE of every image (f32 sums of f16-rounded columns; the kernel
    blends in packed half: tolerance = f16 output + blend rounding)"""
    from s2anet_amd.alignconv import align_conv_forward, pack_weight
    B = 2
    x, anc, w = config2_inputs(B, seed=4321)
    xh = cu(x).half().contiguous(memory_format=torch.channels_last)
    wh = cu(w * 4).half()                                                   # outputs of order 1: f16 rounding visible
    out = align_conv_forward(xh, cu(anc), wh, 8, relu=True)
    out_p = align_conv_forward(xh, cu(anc), pack_weight(wh, torch.float16), 8, relu=True, packed=True, out_channels=256)
    assert torch.equal(out, out_p)                                          # pre-packed filter: same launch
    got = out.float().cpu().numpy()
    xf, wf = xh.float().cpu().numpy(), wh.float().cpu().numpy()
    for b in range(B):
        off = oracle.align_offsets(anc[b].reshape(-1, 5), 128, 128, 8)[None]
        ref = oracle.deform_conv_forward(np.ascontiguousarray(xf[b:b + 1]), off, wf, f16_cols=True, relu=True)
        err = np.abs(got[b:b + 1] - ref)
        assert err.max() < 2e-2 and err.mean() < 1e-3, (b, err.max(), err.mean())
        assert np.abs(ref).max() > 2.0


def test_config1_alignconv_f16_vs_reference_half_semantics():
    """How far is the f16 production kernel from the reference's OWN half instantiation?  Under `.half()` the reference
    casts the offsets to half (models/dcn/deform_conv.py:45-46) and computes h_im / w_im, the bilinear weights and the
    blend in scalar_t = Half (deform_conv_cuda_kernel.cu:83-114,221-228): at P3 a coordinate near 100 has a quantum of
    1/16 px.  k_dcn_patch keeps coordinates and weights in f32 (DESIGN 2 "known deviations").  configs[1] as stated
    (one [1,256,128,128] level, SURVEY 8(d) config-2 anchors) against oracle.deform_conv_forward_half:
      * the deviation is recorded (gpurun_out/half_path_deviation.json -> DESIGN 2) and bounded;
      * it is the reference's coordinate rounding, not this kernel: the f32-coordinate oracle (f16 columns) sits at the
        same distance from the half oracle, and the kernel is within its f16 tolerance of THAT oracle;
      * on a smooth map (what a trained FPN level looks like next to white noise) it is an order of magnitude smaller."""
    import json
    import os
    from s2anet_amd.alignconv import align_conv_forward
    rec = {}
    for name, smooth in (("white_noise", False), ("smooth", True)):
        x, anc, w = config2_inputs(1, seed=99)
        if smooth:                                             # low-pass: 9x9 box filter twice, renormalised to unit variance
            t = torch.from_numpy(x)
            k = torch.ones(256, 1, 9, 9) / 81
            for _ in range(2):
                t = torch.nn.functional.conv2d(t, k, padding=4, groups=256)
            x = (t / t.std()).numpy()
        xh = cu(x).half().contiguous(memory_format=torch.channels_last)
        wh = cu(w * 4).half()
        got = align_conv_forward(xh, cu(anc), wh, 8, relu=True).float().cpu().numpy()
        xf, wf = xh.float().cpu().numpy(), wh.float().cpu().numpy()
        off = oracle.align_offsets(anc[0].reshape(-1, 5), 128, 128, 8)[None]
        ref_half = oracle.deform_conv_forward_half(np.ascontiguousarray(xf), off, wf, relu=True)
        ref_f32c = oracle.deform_conv_forward(np.ascontiguousarray(xf), off, wf, f16_cols=True, relu=True)
        dev_k = np.abs(got - ref_half)
        dev_o = np.abs(ref_f32c - ref_half)
        own = np.abs(got - ref_f32c)
        scale = float(np.abs(ref_half).mean())
        rec[name] = dict(kernel_vs_half_max=float(dev_k.max()), kernel_vs_half_mean=float(dev_k.mean()),
                         f32coord_oracle_vs_half_max=float(dev_o.max()), f32coord_oracle_vs_half_mean=float(dev_o.mean()),
                         kernel_vs_f32coord_oracle_max=float(own.max()), kernel_vs_f32coord_oracle_mean=float(own.mean()),
                         mean_abs_output=scale, max_abs_output=float(np.abs(ref_half).max()))
        assert own.max() < 2e-2 and own.mean() < 1e-3, rec[name]                 # the kernel's own tolerance (as above)
        assert dev_k.mean() <= 1.05 * dev_o.mean() + 1e-3, rec[name]             # nothing beyond the reference's rounding
        assert dev_k.max() <= 1.05 * dev_o.max() + 2e-2, rec[name]
        assert dev_k.mean() < (0.02 if smooth else 0.08) * max(scale, 1e-3), rec[name]
    assert rec["smooth"]["kernel_vs_half_mean"] / rec["smooth"]["mean_abs_output"] < \
        0.5 * rec["white_noise"]["kernel_vs_half_mean"] / rec["white_noise"]["mean_abs_output"]
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/half_path_deviation.json", "w") as f:
        json.dump(rec, f, indent=1)
    print("half-path deviation:", json.dumps(rec))


def test_config1_alignconv_half_coords_mode_reproduces_reference_half_sampling(monkeypatch):
    """S2A_DCN_HALF_COORDS=1: the sampling table of k_dcn_patch rounds the offsets, h_im / w_im, lh / lw / hh / hw and
    the four weights to binary16 exactly where the reference's scalar_t = Half instantiation does
    (models/dcn/deform_conv.py:45-46, deform_conv_cuda_kernel.cu:97-109,221-228).  configs[1] as stated, against
    oracle.deform_conv_forward_half: <= 2e-2 (the f16 tolerance of the f32-coordinate mode against ITS oracle), where the
    default mode sits 0.33 away on white noise.  What stays different by design: the blend is three packed FMAs (one
    rounding each) where the reference rounds product and sum separately (<= 1 half-ulp per corner), and the filter
    contraction accumulates in f32 on the matrix cores."""
    import json
    import os
    from s2anet_amd.alignconv import align_conv_forward
    from s2anet_amd.dcn import deform_conv
    rec = {}
    for name, smooth in (("white_noise", False), ("smooth", True)):
        x, anc, w = config2_inputs(1, seed=99)
        if smooth:
            t = torch.from_numpy(x)
            k = torch.ones(256, 1, 9, 9) / 81
            for _ in range(2):
                t = torch.nn.functional.conv2d(t, k, padding=4, groups=256)
            x = (t / t.std()).numpy()
        xh = cu(x).half().contiguous(memory_format=torch.channels_last)
        wh = cu(w * 4).half()
        xf, wf = xh.float().cpu().numpy(), wh.float().cpu().numpy()
        off = oracle.align_offsets(anc[0].reshape(-1, 5), 128, 128, 8)[None]
        ref_half = oracle.deform_conv_forward_half(np.ascontiguousarray(xf), off, wf, relu=True)
        monkeypatch.delenv("S2A_DCN_HALF_COORDS", raising=False)
        base = align_conv_forward(xh, cu(anc), wh, 8, relu=True).float().cpu().numpy()
        monkeypatch.setenv("S2A_DCN_HALF_COORDS", "1")
        fused = align_conv_forward(xh, cu(anc), wh, 8, relu=True).float().cpu().numpy()          # offsets from the anchors, in-kernel
        # the reference's call shape; its offsets arrive as half values (deform_conv.py:45-46 `offset.type_as(input)`)
        plain = torch.relu(deform_conv(xh, cu(off).half().float(), wh, 1, 1)).float().cpu().numpy()
        monkeypatch.delenv("S2A_DCN_HALF_COORDS")
        again = align_conv_forward(xh, cu(anc), wh, 8, relu=True).float().cpu().numpy()
        assert np.array_equal(base, again)                                  # the switch is per call, the default is untouched
        d_plain, d_fused, d_base = np.abs(plain - ref_half), np.abs(fused - ref_half), np.abs(base - ref_half)
        rec[name] = dict(half_mode_explicit_offsets_max=float(d_plain.max()), half_mode_explicit_offsets_mean=float(d_plain.mean()),
                         half_mode_fused_anchors_max=float(d_fused.max()), half_mode_fused_anchors_mean=float(d_fused.mean()),
                         half_mode_fused_anchors_q9999=float(np.quantile(d_fused, 0.9999)),
                         default_mode_max=float(d_base.max()), default_mode_mean=float(d_base.mean()),
                         mean_abs_output=float(np.abs(ref_half).mean()))
        # same offsets as the oracle: the parity bar
        assert d_plain.max() <= 2e-2 and d_plain.mean() < 1e-3, rec[name]
        # offsets computed in-kernel from the anchors (f32, a last-bit difference from get_offset can cross a binary16
        # rounding boundary: one coordinate quantum for that tap): everywhere else the same bar
        assert np.quantile(d_fused, 0.9999) <= 2e-2 and d_fused.mean() < 1e-3, rec[name]
        assert d_base.mean() > 3 * d_fused.mean(), rec[name]                # the mode really is the reference's rounding
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/half_coords_mode.json", "w") as f:
        json.dump(rec, f, indent=1)
    print("half-coords mode:", json.dumps(rec))


# ------------------------------------------------------------------------------------------------ configs[2]
def _cpu_and_gpu_detectors(dtype, seed=1234, target=2500):
    """the same seeded network twice: the CPU float32 module (oracle/pipeline.py runs it) and the product on the GPU.
    The classifier is calibrated on the CPU (deterministic) and the two calibrated parameters are copied over."""
    from s2anet_amd.detector import BottleNeck, S2ANet, build_synthetic_detector
    torch.manual_seed(seed)
    cpu = S2ANet(num_classes=15).eval()
    for mod in cpu.modules():
        if isinstance(mod, BottleNeck):
            mod.bn3.weight.data.fill_(0.25)
    gpu = build_synthetic_detector(num_classes=15, seed=seed, dtype=dtype, device=dev())
    g = torch.Generator().manual_seed(seed + 7)
    img = torch.randint(0, 256, (1, 3, 1024, 1024), dtype=torch.uint8, generator=g)
    if dtype == torch.float16:      # the CPU module computes in f32 on the f16-rounded parameters of the GPU model
        with torch.no_grad():
            for p in cpu.parameters():
                p.copy_(p.half().float())
    feats, levels = pipeline.forward_chip(cpu, img)
    ncand = pipeline.calibrate_classifier(cpu, levels, target)
    with torch.no_grad():
        gpu.head.odm_cls_head.weight.copy_(cpu.head.odm_cls_head.weight.to(dev(), dtype))
        gpu.head.odm_cls_head.bias.copy_(cpu.head.odm_cls_head.bias.to(dev(), dtype))
    if dtype == torch.float16:      # ... including the calibrated ones
        with torch.no_grad():
            cpu.head.odm_cls_head.weight.copy_(gpu.head.odm_cls_head.weight.float().cpu())
            cpu.head.odm_cls_head.bias.copy_(gpu.head.odm_cls_head.bias.float().cpu())
            for lv in levels:
                c = cpu.head.odm_cls_head(lv["cls_feat"])
                lv["cls"] = c[0].permute(1, 2, 0).reshape(-1, c.shape[1]).numpy().copy()
    return cpu, gpu, img, feats, levels, ncand


def _match(gd, gl, cd, cl, tol_rel=1e-3, tol_score=1e-4):
    """one-to-one matching of GPU detections to CPU detections: same label, box within tol_rel of its size (angle
    1e-3 rad), score within tol_score.  Order-free: detections whose scores differ in the last bits may swap places.
    -> (pairs, unmatched GPU rows, unmatched CPU rows)"""
    used = np.zeros(len(cd), bool)
    pairs, left = [], []
    for i in range(len(gd)):
        cand = np.nonzero((cl == gl[i]) & ~used)[0]
        j = -1
        if cand.size:
            size = np.maximum(cd[cand, 2:4].max(1), 1.0)
            d = np.abs(cd[cand, :4] - gd[i, :4]).max(1) / size
            da = np.abs(cd[cand, 4] - gd[i, 4])
            da = np.minimum(da, np.abs(da - np.float32(np.pi)))         # the angle wraps at the ends of [-pi/4, 3pi/4)
            ok = (d < tol_rel) & (da < 1e-3) & (np.abs(cd[cand, 5] - gd[i, 5]) < tol_score)
            if ok.any():
                j = cand[np.nonzero(ok)[0][np.argmin(d[ok])]]
        if j >= 0:
            used[j] = True
            pairs.append((i, j))
        else:
            left.append(i)
    return pairs, left, np.nonzero(~used)[0].tolist()


@pytest.mark.parametrize("seed", [1234, 77])
def test_config2_detect_f32_vs_cpu_pipeline(seed):
    """BASELINE configs[2] for one 1024 x 1024 chip, float32 (two seeded networks / images): image -> detections on the GPU (detect(): own f32 AlignConv
    on the matrix cores, fused anchor refine, ARF, pooling, decode, on-device segmented ml-NMS; library f32 convolutions
    for the plain layers) against the CPU pipeline (torch CPU convolutions + the oracle's ops, `>` rule and GPU sort
    branch as the reference's CUDA op).  Same number of detections, same labels, boxes within 1e-3 relative to the
    box size, scores within 1e-4 (models/head.py:648-725)."""
    cpu, gpu, img, feats, levels, ncand = _cpu_and_gpu_detectors(torch.float32, seed=seed)
    assert 2000 < ncand < 3000
    dets_c, labels_c, bboxes_c, scores_c = pipeline.postprocess(levels)
    with torch.no_grad():
        d, l, c, ovf = gpu.detect(img.to(dev()), return_overflow=True)
    K = int(c[0])
    assert int(ovf[1]) == 0 and int(ovf[0]) == ncand                       # same candidate set size, nothing dropped
    gd, gl = d[0, :K].cpu().numpy(), l[0, :K].cpu().numpy()
    assert (l[0, K:] == -1).all() and (d[0, K:] == 0).all()
    assert abs(K - len(dets_c)) <= 2 and 500 < K < 2000, (K, len(dets_c))   # below max_per_img: no truncation boundary in play
    pairs, left_g, left_c = _match(gd, gl, dets_c, labels_c.astype(np.int32))
    # Everything must match one to one -- except where float noise legitimately decides otherwise: two overlapping
    # candidates of one class whose scores differ in the 6th digit (the f32 library convolutions are not even
    # run-to-run deterministic) swap roles, so the other one is kept.  At most a handful, and every unmatched GPU detection
    # must be one of the CPU pipeline's own (box, class) CANDIDATES with the same score (nothing invented, nothing moved).
    assert len(left_g) <= 3 and len(left_c) <= 3, (len(left_g), len(left_c))
    rows, cols = np.nonzero(scores_c > 0.05)
    cand = np.concatenate([bboxes_c[rows], scores_c[rows, cols][:, None]], 1)
    for i in left_g:
        same = np.nonzero(cols == gl[i])[0]
        d = np.abs(cand[same, :4] - gd[i, :4]).max(1) / np.maximum(cand[same, 2:4].max(1), 1.0)
        assert same.size and d.min() < 1e-3 and abs(cand[same[np.argmin(d)], 5] - gd[i, 5]) < 1e-4, (i, gd[i])
    assert (np.diff(gd[:, 5]) <= 0).all()                                   # descending score, as the reference returns


def test_config2_detect_f16_stages_vs_cpu_pipeline():
    """the benchmarked path (f16, fused stem, own convolutions, pyramid-packed head) on one 1024 x 1024 chip:
    (A) every stage's dense output against the CPU pipeline in f32 on the same f16-rounded parameters (tolerances of
        ~60 layers of f16 activations), (B) the post-processing chain EXACTLY, each step from the GPU's own inputs:
        top-k selection indices == oracle, scores / boxes of the selected rows, NMS of the GPU's candidates == oracle
        multiclass_nms_rotated (counts, labels, boxes bit for bit) -- the chain the reference runs in
        get_bboxes_single_img (models/head.py:684-725)"""
    from s2anet_amd import pyramid as P
    cpu, gpu, img, feats, levels, ncand = _cpu_and_gpu_detectors(torch.float16)
    imgs = img.to(dev()).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert gpu.backbone.stem_fusable(imgs)
        p = gpu.features_to_pred(imgs, gpu.backbone.forward_u8(imgs, 255.0))
    layout, cls, reg, anc = p.packed
    # (A) dense maps: refined anchors, class logits, box deltas per level
    for li, lv in enumerate(levels):
        H, W = lv["size"]
        a_g = layout.rows(anc, li).view(H * W, 5).cpu().numpy()
        side = 4.0 * lv["stride"]
        da = np.abs(a_g - lv["refined"])
        assert da[:, :4].max() < 1e-3 * side and da[:, 4].max() < 1e-3, (li, da.max(0))   # measured: <= 5e-3 px at side 512
        c_g = layout.level(cls, li, 15)[0].permute(1, 2, 0).reshape(-1, 15).float().cpu().numpy()
        r_g = layout.level(reg, li, 5)[0].permute(1, 2, 0).reshape(-1, 5).float().cpu().numpy()
        ec, er = np.abs(c_g - lv["cls"]), np.abs(r_g - lv["reg"])
        assert ec.mean() < 1e-2 and ec.max() < 0.1, (li, ec.mean(), ec.max())          # logits of std 1.5 (measured: 3e-3 / 1.7e-2)
        assert er.mean() < 2e-5 and er.max() < 1e-4, (li, er.mean(), er.max())          # deltas of std 1e-3 (measured: 2e-6 / 1.2e-5)
    # (B1) candidate selection from the GPU's own maps: exact indices; f16 sigmoid within one f16 ulp; boxes 1e-4
    bb, sc, sel = P.candidates(layout, cls, reg, anc, 15, gpu.head.max_before_nms_per_level)
    glv = []
    for li in range(len(layout.sizes)):
        H, W = layout.sizes[li]
        glv.append(dict(cls=layout.level(cls, li, 15)[0].permute(1, 2, 0).reshape(-1, 15).float().cpu().numpy(),
                        reg=layout.level(reg, li, 5)[0].permute(1, 2, 0).reshape(-1, 5).float().cpu().numpy(),
                        refined=layout.rows(anc, li).view(H * W, 5).cpu().numpy()))
    s_o, d_o, a_o, rows_o, lev_o = pipeline.select_candidates(glv, 2000, half_scores=True)
    packed_rows = np.array([layout.pix0[l] for l in lev_o]) + rows_o
    assert np.array_equal(sel[0].cpu().numpy(), packed_rows)
    sc_g, bb_g = sc[0].cpu().numpy(), bb[0].cpu().numpy()
    ulp = np.maximum(np.abs(s_o), 2.0 ** -14) * 2.0 ** -10
    assert (np.abs(sc_g - s_o) <= ulp).all()
    b_o = oracle.delta2bbox_rotated(a_o, d_o)
    assert np.allclose(bb_g, b_o, rtol=1e-4, atol=1e-3)
    # (B2) NMS of exactly the GPU's candidates: the oracle's multiclass_nms_rotated, `>` rule, GPU sort branch
    with torch.no_grad():
        d, l, c, ovf = gpu.head.get_bboxes_batched(p, return_overflow=True)
    K = int(c[0])
    dets_o, labels_o = oracle.multiclass_nms_rotated(bb_g, sc_g, 0.05, 0.5, 2000)
    assert int(ovf[0]) == int((sc_g > 0.05).sum()) and int(ovf[1]) == 0
    assert K == len(dets_o) and K > 300, (K, len(dets_o))
    gd, gl = d[0, :K].cpu().numpy(), l[0, :K].cpu().numpy()
    # f16-rounded scores tie: equal-score detections may come in either order -> compare as sorted row sets
    key_g = np.lexsort((gd[:, 0], gd[:, 1], gl, -gd[:, 5]))
    key_o = np.lexsort((dets_o[:, 0], dets_o[:, 1], labels_o, -dets_o[:, 5]))
    assert np.array_equal(gl[key_g], labels_o[key_o].astype(np.int32))
    assert np.array_equal(gd[key_g].view(np.uint32), dets_o[key_o].view(np.uint32))
    # and detect() on the uint8 batch is that same chain (a second run of the trunk: its two library convolutions are
    # not run-to-run deterministic, so a handful of near-threshold candidates may differ)
    with torch.no_grad():
        d2, l2, c2 = gpu.detect(imgs)
    assert abs(int(c2[0]) - K) <= max(3, K // 100), (int(c2[0]), K)


def test_config2_batch8_postprocessing_every_image_vs_oracle():
    """BASELINE configs[2] at the benchmarked batch: 8 DISTINCT 1024 x 1024 chips through the f16 detector (fused stem,
    174 592-row pyramid, image x class segments, one segmented NMS call).  From the GPU's own dense maps, for EVERY image:
    top-k selection indices == oracle, scores within one f16 ulp, decoded boxes 1e-4, and the NMS of the GPU's own
    candidates == oracle.multiclass_nms_rotated bit for bit (models/head.py:684-725, utils/bbox_nms_rotated.py:5-64).
    Then batched == per-chip: image b's rows re-packed as a batch-1 pyramid give the same detections bit for bit."""
    from s2anet_amd import pyramid as P
    from s2anet_amd.head import PyramidPred
    from s2anet_amd.pyramid import PyramidLayout
    cpu, gpu, img, feats, levels, ncand = _cpu_and_gpu_detectors(torch.float16)
    g = torch.Generator().manual_seed(2024)
    B = 8
    chips = [img] + [torch.randint(0, 256, (1, 3, 1024, 1024), dtype=torch.uint8, generator=g) for _ in range(B - 1)]
    # distinct content, not 8 draws of one distribution: brightness ramps / flat patches change the per-chip candidate counts
    for b in range(1, B):
        chips[b][:, :, : 128 * b] //= (b + 1)
    imgs = torch.cat(chips).to(dev()).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert gpu.backbone.stem_fusable(imgs)
        p = gpu.features_to_pred(imgs, gpu.backbone.forward_u8(imgs, 255.0))
        layout, cls, reg, anc = p.packed
        assert layout.batch == B and layout.pixels == 174592
        bb, sc, sel = P.candidates(layout, cls, reg, anc, 15, gpu.head.max_before_nms_per_level)
        d, l, c, ovf = gpu.head.get_bboxes_batched(p, return_overflow=True)
    assert int(ovf[1]) == 0 and int(ovf[0]) == int((sc > 0.05).sum())
    counts = c.cpu().numpy()
    assert counts.sum() > 2000 and len(set(counts.tolist())) > 4, counts     # really different chips
    sizes = layout.sizes
    for b in range(B):
        glv = []
        for li, (H, W) in enumerate(sizes):
            glv.append(dict(cls=layout.level(cls, li, 15)[b].permute(1, 2, 0).reshape(-1, 15).float().cpu().numpy(),
                            reg=layout.level(reg, li, 5)[b].permute(1, 2, 0).reshape(-1, 5).float().cpu().numpy(),
                            refined=layout.rows(anc, li).view(B, H * W, 5)[b].cpu().numpy()))
        s_o, d_o, a_o, rows_o, lev_o = pipeline.select_candidates(glv, 2000, half_scores=True)
        packed_rows = np.array([layout.pix0[lv] + b * sizes[lv][0] * sizes[lv][1] for lv in lev_o]) + rows_o
        assert np.array_equal(sel[b].cpu().numpy(), packed_rows), b
        sc_g, bb_g = sc[b].cpu().numpy(), bb[b].cpu().numpy()
        ulp = np.maximum(np.abs(s_o), 2.0 ** -14) * 2.0 ** -10
        assert (np.abs(sc_g - s_o) <= ulp).all(), b
        assert np.allclose(bb_g, oracle.delta2bbox_rotated(a_o, d_o), rtol=1e-4, atol=1e-3), b
        dets_o, labels_o = oracle.multiclass_nms_rotated(bb_g, sc_g, 0.05, 0.5, 2000)
        K = int(counts[b])
        assert K == len(dets_o), (b, K, len(dets_o))
        gd, gl = d[b, :K].cpu().numpy(), l[b, :K].cpu().numpy()
        assert (l[b, K:] == -1).all() and (d[b, K:] == 0).all()
        key_g = np.lexsort((gd[:, 0], gd[:, 1], gl, -gd[:, 5]))             # f16-rounded scores tie: compare as sorted row sets
        key_o = np.lexsort((dets_o[:, 0], dets_o[:, 1], labels_o, -dets_o[:, 5]))
        assert np.array_equal(gl[key_g], labels_o[key_o].astype(np.int32)), b
        assert np.array_equal(gd[key_g].view(np.uint32), dets_o[key_o].view(np.uint32)), b
    # batched == per-chip (same maps, batch-1 pyramid)
    one = PyramidLayout(1, sizes, layout.strides)
    for b in range(B):
        def rows_of(buf):
            return torch.cat([layout.rows(buf, li).view(B, sizes[li][0] * sizes[li][1], buf.shape[1])[b]
                              for li in range(len(sizes))]).contiguous()
        p1 = PyramidPred(one, rows_of(cls), rows_of(reg), rows_of(anc), [], [], [], [], [])
        with torch.no_grad():
            d1, l1, c1 = gpu.head.get_bboxes_batched(p1)
        assert int(c1[0]) == int(counts[b]), b
        assert torch.equal(d1[0], d[b]) and torch.equal(l1[0], l[b]), b


def test_candidate_cap_overflow_is_reported():
    """the reference never drops a candidate (utils/bbox_nms_rotated.py:29-40); a static cap below the candidate count
    must say so: overflow = [found, dropped]; with a sufficient cap dropped == 0 and the result equals the uncapped one"""
    from s2anet_amd.rotated import batched_multiclass_nms_rotated
    rng = np.random.default_rng(5)
    B, n, C = 2, 600, 15
    from conftest import rand_rboxes
    bb = cu(np.stack([rand_rboxes(rng, n, span=400) for _ in range(B)]))
    sc = cu(rng.uniform(0, 0.2, (B, n, C)).astype(np.float32))
    found = int((sc > 0.05).sum())
    assert found > 10000
    d0, l0, c0, o0 = batched_multiclass_nms_rotated(bb, sc, 0.05, 0.5, 2000, None, return_overflow=True)
    assert o0.cpu().tolist() == [found, 0]
    d1, l1, c1, o1 = batched_multiclass_nms_rotated(bb, sc, 0.05, 0.5, 2000, found, return_overflow=True)
    assert o1.cpu().tolist() == [found, 0] and torch.equal(d1, d0) and torch.equal(l1, l0) and torch.equal(c1, c0)
    cap = found // 3
    d2, l2, c2, o2 = batched_multiclass_nms_rotated(bb, sc, 0.05, 0.5, 2000, cap, return_overflow=True)
    assert o2.cpu().tolist() == [found, found - cap]
    assert int(c2.sum()) < int(c0.sum())                                    # rows really were cut
    # detect() hands the same pair through
    from s2anet_amd.detector import build_synthetic_detector
    m = build_synthetic_detector(device=dev())
    m.head.odm_cls_head.bias.data.fill_(-1.0)                               # sigmoid(-1) = 0.27 > 0.05: every (box, class) is a candidate
    img = torch.randint(0, 256, (1, 3, 256, 256), dtype=torch.uint8, device=dev()).contiguous(memory_format=torch.channels_last)
    d, l, c, o = m.detect(img, max_candidates=1000, return_overflow=True)
    assert int(o[0]) > 1000 and int(o[1]) == int(o[0]) - 1000
    assert len(m.detect(img, max_candidates=1000)) == 3                     # default return shape unchanged


def test_detect_hip_graph_replay_equals_eager():
    """SURVEY 8(f) row 4 (models/head.py:296-348 + :648-725 as one captured HIP graph): capture the head + post-processing
    of detect() (own kernels only: bit-reproducible) on a small pyramid, replay it on two further inputs written into
    the captured input buffer, and compare with eager execution bit for bit; then the whole detect() captured and
    replayed on a changed uint8 batch (trunk included) against eager within the trunk's run-to-run tolerance"""
    from s2anet_amd.detector import build_synthetic_detector
    from s2anet_amd.pyramid import PyramidLayout
    m = build_synthetic_detector(device=dev())
    m.head.odm_cls_head.bias.data.fill_(-2.0)
    m.head.odm_cls_head.weight.data.mul_(20.0)
    layout = PyramidLayout(2, [(40, 48), (20, 24), (10, 12), (5, 6), (3, 3)], (8, 16, 32, 64, 128))
    g = torch.Generator().manual_seed(11)
    feats = [torch.randn(layout.pixels, 256, generator=g).to(dev()).half() for _ in range(3)]

    def run(x):
        return m.head.get_bboxes_batched(m.head.forward_pyramid(layout, x), max_candidates=50000, return_overflow=True)
    with torch.no_grad():
        eager = [tuple(t.clone() for t in run(x)) for x in feats]
        static_x = feats[0].clone()
        side = torch.cuda.Stream(device=dev())
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm the side stream's workspaces before capturing on it
            run(static_x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            static_out = run(static_x)
        for k in (1, 2, 0, 2):
            static_x.copy_(feats[k])
            graph.replay()
            torch.cuda.synchronize()
            for got, ref in zip(static_out, eager[k]):
                assert torch.equal(got, ref), k
    assert int(eager[0][2].sum()) > 100 and not torch.equal(eager[0][0], eager[1][0])
    # whole detect(): uint8 batch in a static buffer
    imgs = [torch.randint(0, 256, (2, 3, 256, 320), dtype=torch.uint8, generator=g).to(dev()).contiguous(memory_format=torch.channels_last)
            for _ in range(2)]
    with torch.no_grad():
        ref = [tuple(t.clone() for t in m.detect(i)) for i in imgs]
        static_img = imgs[0].clone(memory_format=torch.channels_last)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.detect(static_img)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=side):
            out = m.detect(static_img)
        for k in (1, 0):
            static_img.copy_(imgs[k])
            g2.replay()
            torch.cuda.synchronize()
            assert abs(int(out[2].sum()) - int(ref[k][2].sum())) <= max(2, int(ref[k][2].sum()) // 50), k
