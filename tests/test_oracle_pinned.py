"""CPU: the oracle (oracle/) against the golden vectors generated from the reference
(tests/golden/make_golden.py) and, when oracle/_ref is present, against the reference's
own CPU ops live.  Bit-exact for IoU (both sort branches), NMS keep lists and ARF."""
import numpy as np
import pytest

import oracle
from oracle import ref
from conftest import golden, rand_rboxes, distinct_scores


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_iou_matches_reference_cpu_bitexact():
    g = golden("iou_256.npz")
    o = oracle.box_iou_rotated(g["boxes1"], g["boxes2"], sort_mode=oracle.SORT_CPU)
    assert np.array_equal(bits(o), bits(g["iou_ref_cpu"]))


def test_iou_matches_reference_gpubranch_bitexact():
    g = golden("iou_256.npz")
    o = oracle.box_iou_rotated(g["boxes1"], g["boxes2"], sort_mode=oracle.SORT_GPU)
    assert np.array_equal(bits(o), bits(g["iou_ref_gpubranch"]))


def test_iou_cull_shortcut_is_exact():
    g = golden("iou_256.npz")
    for mode in (oracle.SORT_CPU, oracle.SORT_GPU):
        a = oracle.box_iou_rotated(g["boxes1"], g["boxes2"], sort_mode=mode, cull=False)
        b = oracle.box_iou_rotated(g["boxes1"], g["boxes2"], sort_mode=mode, cull=True)
        assert np.array_equal(bits(a), bits(b))


def test_known_answers():
    # the reference's own commented example: unit squares offset by 0.5 -> 1/7 (polyiou.cpp:130-137)
    sq1 = np.array([[0.5, 0.5, 1, 1, 0]], np.float32)
    sq2 = np.array([[1.0, 1.0, 1, 1, 0]], np.float32)
    assert abs(oracle.box_iou_rotated(sq1, sq2)[0, 0] - 1 / 7) < 1e-6
    assert abs(oracle.polyiou([0, 0, 1, 0, 1, 1, 0, 1], [0.5, 0.5, 1.5, 0.5, 1.5, 1.5, 0.5, 1.5])[0] - 1 / 7) < 1e-12
    # square vs itself rotated 45 deg -> 0.7071
    a = np.array([[0, 0, 2, 2, 0]], np.float32)
    b = np.array([[0, 0, 2, 2, np.pi / 4]], np.float32)
    assert abs(oracle.box_iou_rotated(a, b)[0, 0] - 0.707107) < 1e-5
    # degenerate
    z = np.array([[0, 0, 1e-8, 1e-8, 0]], np.float32)
    assert oracle.box_iou_rotated(a, z)[0, 0] == 0.0
    assert oracle.box_iou_rotated(np.zeros((0, 5), np.float32), a).shape == (0, 1)


def test_polyiou_matches_reference_and_rotated_iou():
    g = golden("iou_256.npz")
    P1, P2 = oracle.rboxes_to_polys(g["boxes1"]), oracle.rboxes_to_polys(g["boxes2"])
    p = oracle.polyiou(P1[g["poly_i"]], P2[g["poly_j"]])
    assert np.array_equal(p, g["poly_iou"])          # f64 bit-exact vs reference polyiou
    r = g["iou_ref_cpu"][g["poly_i"], g["poly_j"]]
    big = np.maximum(g["boxes1"][g["poly_i"], 2:4].max(1), g["boxes2"][g["poly_j"], 2:4].max(1)) < 150
    assert np.abs(p[big] - r[big]).max() < 1e-4      # the two algorithms agree


@pytest.mark.parametrize("thr", [0.1, 0.5])
def test_nms_keep_matches_reference(thr):
    g = golden("nms_2k.npz")
    d, s, lab = g["dets"], g["scores"], g["labels"]
    k = oracle.ml_nms_rotated(d, s, lab, thr, rule=oracle.RULE_GE, sort_mode=oracle.SORT_CPU)
    assert np.array_equal(k, g[f"ml_keep_ge_{thr}"])
    k = oracle.nms_rotated(d, s, thr, rule=oracle.RULE_GE, sort_mode=oracle.SORT_CPU)
    assert np.array_equal(k, g[f"sc_keep_ge_{thr}"])
    # GPU rule + GPU sort branch (restated): recorded keep lists are reproducible, and the
    # exact-cull fast path of the oracle gives the same list
    for cull in (False, True):
        k = oracle.ml_nms_rotated(d, s, lab, thr, rule=oracle.RULE_GT, sort_mode=oracle.SORT_GPU, cull=cull)
        assert np.array_equal(k, g[f"ml_keep_gt_{thr}"])
        k = oracle.nms_rotated(d, s, thr, rule=oracle.RULE_GT, sort_mode=oracle.SORT_GPU, cull=cull)
        assert np.array_equal(k, g[f"sc_keep_gt_{thr}"])


def test_nms_four_box_case():
    g = golden("nms_2k.npz")
    assert list(g["ml_keep4"]) == [0, 2, 3] and list(g["sc_keep4"]) == [0, 3]
    assert list(oracle.ml_nms_rotated(g["d4"], g["s4"], g["l4"], 0.5)) == [0, 2, 3]
    assert list(oracle.nms_rotated(g["d4"], g["s4"], 0.5)) == [0, 3]
    # rule difference: identical boxes have iou == 1.0; thr=1.0 separates > from >=
    assert list(oracle.nms_rotated(g["d4"], g["s4"], 1.0, rule=oracle.RULE_GT)) == [0, 1, 2, 3]
    assert list(oracle.nms_rotated(g["d4"], g["s4"], 1.0, rule=oracle.RULE_GE)) == [0, 3]
    assert oracle.nms_rotated(np.zeros((0, 5), np.float32), np.zeros(0, np.float32), 0.5).shape == (0,)


def test_arf_matches_reference_small_and_definition_large():
    g = golden("arf_small.npz")
    for tag in ("s1", "s8"):
        assert np.array_equal(oracle.arf_forward(g[f"w_{tag}"], g[f"idx_{tag}"]), g[f"out_{tag}"])
    # production shape: every output element written exactly once (idx[:,k] is a permutation)
    rng = np.random.default_rng(0)
    w = rng.standard_normal((32, 256, 1, 3, 3)).astype(np.float32)
    idx = oracle.arf_indices(1, 8, 3)
    out = oracle.arf_forward(w, idx)
    assert out.shape == (256, 256, 3, 3)
    assert np.array_equal(np.sort(out.reshape(32, 8, 256, 9), -1),
                          np.sort(np.broadcast_to(w.reshape(32, 1, 256, 9), (32, 8, 256, 9)), -1))
    assert np.array_equal(out.reshape(32, 8, 256, 9)[:, 0], w.reshape(32, 256, 9))  # rotation 0 = identity


def test_arf_backward_matches_reference():
    g = golden("arf_backward_small.npz")
    for tag in ("s1", "s8"):
        assert np.array_equal(oracle.arf_backward(g[f"idx_{tag}"], g[f"gout_{tag}"]), g[f"gin_{tag}"])
    # adjoint identity at production shape: <arf(w), g> == <w, arf_backward(g)>
    rng = np.random.default_rng(2)
    idx = oracle.arf_indices(1, 8, 3)
    w = rng.standard_normal((32, 256, 1, 3, 3)).astype(np.float32)
    gy = rng.standard_normal((256, 256, 3, 3)).astype(np.float32)
    lhs = float((oracle.arf_forward(w, idx).astype(np.float64) * gy).sum())
    rhs = float((w.astype(np.float64) * oracle.arf_backward(idx, gy)).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def test_rie_matches_reference():
    """RotationInvariantEncoding forward / backward restated in numpy == the reference's CPU op (ties: first index;
    the all-equal group keeps direction 0)"""
    g = golden("rie_small.npz")
    for tag in ("a", "b"):
        n = int(g[f"n_{tag}"])
        d, al = oracle.rie_forward(g[f"f_{tag}"], n)
        assert np.array_equal(d, g[f"dir_{tag}"]) and np.array_equal(al, g[f"aligned_{tag}"])
        assert np.array_equal(oracle.rie_backward(g[f"dir_{tag}"], g[f"gout_{tag}"], n), g[f"gin_{tag}"])
        # backward is the inverse rotation of forward
        assert np.array_equal(oracle.rie_backward(d, al, n), g[f"f_{tag}"])


def test_dcn_backward_oracle_matches_autograd_of_independent_formulation():
    """parity unpinned against the reference itself (CUDA only); pinned to autograd of the torch formulation"""
    g = golden("dcn_backward_small.npz")
    gx, goff, gw = oracle.deform_conv_backward(g["x"], g["offset"], g["weight"], g["grad_out"])
    for got, ref in ((gx, g["grad_input"]), (goff, g["grad_offset"]), (gw, g["grad_weight"])):
        assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


VOC_CASES = (("default", {}), ("voc07", dict(use_07_metric=True)), ("hard", dict(is_filter_difficult=False)),
             ("thr07", dict(ovthresh=0.7)))


def test_voc_eval_matches_reference_script():
    g = golden("voc_eval.npz")
    a = (g["det_polys"], g["det_scores"], g["det_image"], g["gt_polys"], g["gt_image"], g["gt_difficult"], int(g["num_images"]))
    for tag, kw in VOC_CASES:
        rec, prec, ap, _, _ = oracle.voc_eval_arrays(*a, **kw)
        assert np.array_equal(rec, g["rec_" + tag]) and np.array_equal(prec, g["prec_" + tag]), tag
        assert ap == float(g["ap_" + tag]), tag


ASSIGN_CASES = (("default", {}), ("first", dict(gt_max_assign_all=False)),
                ("thr", dict(pos_iou_thr=0.3, neg_iou_thr=0.2, min_pos_iou_thr=0.1)), ("nofilter", dict(filter_invalid_anchors=False)))


def test_assign_labels_matches_reference_python():
    g = golden("assign_labels.npz")
    for tag, kw in ASSIGN_CASES:
        got = oracle.assign_labels(g["anchors"], g["gts"], sort_mode=oracle.SORT_CPU, **kw)
        assert np.array_equal(got, g["assign_" + tag]), tag
    assert np.array_equal(oracle.assign_labels(g["anchors"], np.zeros((0, 5), np.float32)), g["assign_empty"])


def test_merge_nms_poly_matches_reference_script():
    g = golden("merge_nms_poly.npz")
    for thr in (0.1, 0.5):
        assert np.array_equal(oracle.nms_poly(g["dets"], thr), g[f"keep_{thr}"])


def test_head_glue_matches_reference_python():
    g = golden("head_glue.npz")
    assert np.array_equal(oracle.grid_anchors(12, 20, 8), g["anchors_s8"])
    assert np.array_equal(oracle.grid_anchors(5, 7, 32), g["anchors_s32"])
    assert list(g["anchors_s8"][0]) == [3.5, 3.5, 32, 32, 0]
    for key, clip in (("dec_clip_fam", 1e-6), ("dec_clip_odm", 16 / 1000)):
        o = oracle.delta2bbox_rotated(g["dec_anchors"], g["dec_deltas"], clip)
        assert np.allclose(o, g[key], rtol=1e-5, atol=1e-4), key
    off = oracle.align_offsets(g["off_anchors"], 12, 20, 8)
    assert off.shape == (18, 12, 20)
    assert np.allclose(off, g["off_s8"], rtol=1e-5, atol=1e-4)
    assert np.array_equal(oracle.arf_indices(1, 8), g["orconv_indices_1_8"])
    assert np.array_equal(oracle.arf_indices(8, 8), g["orconv_indices_8_8"])
    det, lab = oracle.multiclass_nms_rotated(g["mc_bboxes"], g["mc_scores"], 0.05, 0.5, 300,
                                             rule=oracle.RULE_GE, sort_mode=oracle.SORT_CPU)
    assert np.array_equal(det, g["mc_det"]) and np.array_equal(lab, g["mc_labels"])
    assert tuple(g["mc_empty_det_shape"]) == (0, 6)


def test_deform_conv_restatement():
    g = golden("dcn_small.npz")
    o = oracle.deform_conv_forward(g["x"], g["offset"], g["weight"])
    assert np.allclose(o, g["out_torch"], rtol=1e-5, atol=1e-5)
    # zero offsets == plain convolution (identity named in SURVEY 8c)
    import torch
    import torch.nn.functional as F
    x, w = torch.from_numpy(g["x"]), torch.from_numpy(g["weight"])
    z = oracle.deform_conv_forward(g["x"], np.zeros_like(g["offset"]), g["weight"])
    assert np.allclose(z, F.conv2d(x, w, padding=1).numpy(), rtol=1e-5, atol=1e-5)
    # groups / deformable groups / stride / dilation plumbing against grouped conv2d
    rng = np.random.default_rng(3)
    x2 = rng.standard_normal((1, 8, 10, 9)).astype(np.float32)
    w2 = rng.standard_normal((6, 4, 3, 3)).astype(np.float32)
    off0 = np.zeros((1, 2 * 2 * 9, 4, 4), np.float32)
    z2 = oracle.deform_conv_forward(x2, off0, w2, stride=(2, 2), padding=(1, 1), dilation=(2, 2),
                                    groups=2, deformable_groups=2)
    assert np.allclose(z2, F.conv2d(torch.from_numpy(x2), torch.from_numpy(w2), stride=2, padding=1,
                                    dilation=2, groups=2).numpy(), rtol=1e-5, atol=1e-5)


def test_round_f16_helper():
    import torch
    v = np.random.default_rng(0).standard_normal(2000).astype(np.float32) * 100
    v = np.concatenate([v, np.float32([0, 1e-8, 6e-5, 65504, 65519, 70000, -3e-6])])
    exp = torch.from_numpy(v).half().float().numpy()
    got = np.array([oracle.lib().orc_round_f16(float(t)) for t in v], np.float32)
    assert np.array_equal(got, exp)


@pytest.mark.skipif(ref.box_iou_rotated() is None, reason="oracle/_ref not built")
def test_live_reference_ops_agree_on_fresh_inputs(rng):
    import torch
    torch.set_num_threads(1)
    b1, b2 = rand_rboxes(rng, 150, span=200), rand_rboxes(rng, 170, span=200)
    r = ref.box_iou_rotated()(torch.from_numpy(b1), torch.from_numpy(b2)).numpy()
    assert np.array_equal(bits(r), bits(oracle.box_iou_rotated(b1, b2, sort_mode=oracle.SORT_CPU)))
    n = 800
    d, s = rand_rboxes(rng, n, span=250), distinct_scores(rng, n)
    lab = rng.integers(0, 15, n).astype(np.float32)
    rk = ref.ml_nms_rotated()(torch.from_numpy(d), torch.from_numpy(s), torch.from_numpy(lab), 0.3).numpy()
    assert np.array_equal(rk, oracle.ml_nms_rotated(d, s, lab, 0.3, rule=oracle.RULE_GE, sort_mode=oracle.SORT_CPU))


def test_nms_f64_matches_reference_cpu_double_dispatch():
    """the NMS ops dispatch on the dtype of `dets` (AT_DISPATCH_FLOATING_TYPES, nms_rotated_cpu.cpp:66): on float64 boxes
    the reference evaluates single_box_iou_rotated<double>.  Fixture from the reference's CPU ops on data where 136
    keep decisions differ between float32 and float64 arithmetic (tests/golden/make_golden.py:gen_nms_f64)."""
    g = golden("nms_f64.npz")
    d, s, l, thr = g["dets"], g["scores"], g["labels"], float(g["thr"])
    assert d.dtype == np.float64
    k_ml = oracle.nms_rotated_f64(d, s, thr, labels=l, rule=oracle.RULE_GE, sort_mode=oracle.SORT_CPU)
    k_sc = oracle.nms_rotated_f64(d, s, thr, rule=oracle.RULE_GE, sort_mode=oracle.SORT_CPU)
    assert np.array_equal(k_ml, g["ml_keep_f64"]) and np.array_equal(k_sc, g["sc_keep_f64"])
    assert len(set(g["ml_keep_f64"].tolist()) ^ set(g["ml_keep_f32"].tolist())) > 50      # the fixture has teeth
    k32 = oracle.ml_nms_rotated(d.astype(np.float32), s.astype(np.float32), l.astype(np.float32), thr, rule=oracle.RULE_GE,
                                sort_mode=oracle.SORT_CPU)
    assert not np.array_equal(k32, g["ml_keep_f64"])                                        # ... float32 arithmetic fails it
    ref_ml = ref.ml_nms_rotated()
    if ref_ml is not None:                      # live: fresh double inputs through the reference's own CPU op
        import torch
        rng = np.random.default_rng(5)
        dd = rand_rboxes(rng, 500, span=260).astype(np.float64)
        ss = distinct_scores(rng, 500).astype(np.float64)
        ll = rng.integers(0, 4, 500).astype(np.float64)
        want = ref_ml(torch.from_numpy(dd), torch.from_numpy(ss), torch.from_numpy(ll), 0.3).numpy()
        assert np.array_equal(oracle.nms_rotated_f64(dd, ss, 0.3, labels=ll, rule=oracle.RULE_GE, sort_mode=oracle.SORT_CPU), want)


def test_deform_conv_restatement_vs_torch_grid_sample(rng):
    """a SECOND independent formulation of the deformable convolution: torch's own bilinear sampler,
    F.grid_sample(align_corners=True, padding_mode='zeros'), fed with the sampling positions p + tap + offset, then an
    einsum with the filter.  Zero padding of grid_sample == the reference's dropped corners, and its (-1, H) validity band
    == deform_conv_cuda_kernel.cu:228 (a corner at -1 or H contributes zero either way).  Neither this nor the 4-corner
    gather of tests/golden/make_golden.py is the reference -- the op stays "parity unpinned" -- but the oracle now agrees
    with two formulations that share no code, on offsets that leave the image, sit on the border band and are wild."""
    import torch
    import torch.nn.functional as F
    B, C, H, W, O = 2, 6, 11, 13, 5
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((O, C, 3, 3)) * 0.2).astype(np.float32)
    off = (rng.standard_normal((B, 18, H, W)) * 2.5).astype(np.float32)
    off[0, :, 0, :] = -1.4                                   # leaves the image at the top
    off[1, :, :, -1] = 0.9999                                # the (W-1, W) band on the right
    off[0, :, 5, 5] = 40.0
    xt, offt = torch.from_numpy(x).double(), torch.from_numpy(off).double()
    ys = torch.arange(H, dtype=torch.float64).view(1, H, 1)
    xs = torch.arange(W, dtype=torch.float64).view(1, 1, W)
    cols = []
    for t in range(9):
        ky, kx = t // 3, t % 3
        py = ys - 1 + ky + offt[:, 2 * t]                    # [B,H,W] sampling rows / columns in pixels
        px = xs - 1 + kx + offt[:, 2 * t + 1]
        grid = torch.stack([2 * px / (W - 1) - 1, 2 * py / (H - 1) - 1], -1)      # align_corners=True normalisation
        cols.append(F.grid_sample(xt, grid, mode="bilinear", padding_mode="zeros", align_corners=True))
    col = torch.stack(cols, 2)                               # [B,C,9,H,W]
    want = torch.einsum("ock,bckhw->bohw", torch.from_numpy(w).double().view(O, C, 9), col).float().numpy()
    got = oracle.deform_conv_forward(x, off, w)
    assert np.abs(got - want).max() < 2e-5, float(np.abs(got - want).max())


def test_f64_deform_conv_restatement_agrees_with_the_f32_oracle():
    """oracle.deform_conv_forward_f64 (numpy, float64) is the checker of the library's float64 instantiation
    (tests/test_gpu_ops.py); here it is tied to the C++ restatement (float32) on general geometry: same algorithm, float noise"""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 4, 7, 9)).astype(np.float32)
    off = (rng.standard_normal((2, 18, 7, 9)) * 1.5).astype(np.float32)
    w = rng.standard_normal((6, 4, 3, 3)).astype(np.float32)
    assert np.abs(oracle.deform_conv_forward(x, off, w) - oracle.deform_conv_forward_f64(x, off, w)).max() < 2e-5
    x = rng.standard_normal((1, 4, 9, 11)).astype(np.float32)
    w = rng.standard_normal((6, 2, 3, 3)).astype(np.float32)
    off = rng.standard_normal((1, 36, 4, 5)).astype(np.float32)
    kw = dict(stride=(2, 2), padding=(1, 1), dilation=(2, 2), groups=2, deformable_groups=2)
    assert np.abs(oracle.deform_conv_forward(x, off, w, **kw) - oracle.deform_conv_forward_f64(x, off, w, **kw)).max() < 2e-5
