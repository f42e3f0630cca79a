"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle on the same
seeded inputs and against the committed golden vectors.  Bit-exact for IoU values (GPU sort
branch) and NMS keep indices; 1e-4 for AlignConv floats (f32)."""
import numpy as np
import pytest
import torch

import oracle
from conftest import golden, rand_rboxes, distinct_scores

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return t if dtype is None else t.to(dtype)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ------------------------------------------------------------------ rotated IoU
def test_iou_golden_bitexact():
    import s2anet_amd as S
    g = golden("iou_256.npz")
    out = S.box_iou_rotated(cu(g["boxes1"]), cu(g["boxes2"])).cpu().numpy()
    ref = g["iou_ref_gpubranch"]
    neq = bits(out) != bits(ref)
    assert neq.sum() == 0, (neq.sum(), np.abs(out - ref).max())
    assert np.abs(out - g["iou_ref_cpu"]).max() < 1e-4   # vs the reference CPU op: north-star tolerance


def test_iou_random_vs_oracle(rng):
    import s2anet_amd as S
    b1, b2 = rand_rboxes(rng, 1111, span=400), rand_rboxes(rng, 777, span=400)
    out = S.box_iou_rotated(cu(b1), cu(b2)).cpu().numpy()
    ref = oracle.box_iou_rotated(b1, b2, sort_mode=oracle.SORT_GPU, cull=True)
    assert out.shape == (1111, 777)
    assert (bits(out) != bits(ref)).sum() == 0
    assert (out > 0).mean() > 0.01


def test_iou_pairs_and_edges(rng):
    import s2anet_amd as S
    from s2anet_amd.rotated import box_iou_rotated_pairs
    a, b = rand_rboxes(rng, 5000, span=150), rand_rboxes(rng, 5000, span=150)
    out = box_iou_rotated_pairs(cu(a), cu(b)).cpu().numpy()
    ref = oracle.iou_pairs(a, b, sort_mode=oracle.SORT_GPU)
    assert (bits(out) != bits(ref)).sum() == 0
    e = torch.zeros((0, 5), device=dev())
    assert S.box_iou_rotated(e, cu(a[:3])).shape == (0, 3)
    assert S.box_iou_rotated(cu(a[:3]), e).shape == (3, 0)
    sq = cu(np.array([[0.5, 0.5, 1, 1, 0]], np.float32))
    sq2 = cu(np.array([[1.0, 1.0, 1, 1, 0]], np.float32))
    assert abs(S.box_iou_rotated(sq, sq2).item() - 1 / 7) < 1e-6
    # f64 / non-contiguous inputs are converted, not mis-read
    assert abs(S.box_iou_rotated(sq.double(), sq2).item() - 1 / 7) < 1e-6


def test_iou_dense_overlap_chunks(rng):
    """every pair overlapping: the pair list is as long as the matrix"""
    import s2anet_amd as S
    b1, b2 = rand_rboxes(rng, 300, span=20, lo=30, hi=60), rand_rboxes(rng, 260, span=20, lo=30, hi=60)
    out = S.box_iou_rotated(cu(b1), cu(b2)).cpu().numpy()
    ref = oracle.box_iou_rotated(b1, b2, sort_mode=oracle.SORT_GPU)
    assert (out > 0).mean() > 0.95
    assert (bits(out) != bits(ref)).sum() == 0


def test_iou_dense_overlap_wide(rng):
    """every pair of a 1200 x 2100 problem overlapping: full 1024-column workgroups whose circle- and SAT-survivor
    queues fill and drain many times per tile (queue capacity / flush discipline of k_iou_cull)"""
    import s2anet_amd as S
    b1, b2 = rand_rboxes(rng, 1200, span=25, lo=30, hi=60), rand_rboxes(rng, 2100, span=25, lo=30, hi=60)
    out = S.box_iou_rotated(cu(b1), cu(b2)).cpu().numpy()
    ref = oracle.box_iou_rotated(b1, b2, sort_mode=oracle.SORT_GPU)
    assert (out > 0).mean() > 0.95
    assert (bits(out) != bits(ref)).sum() == 0
    # half dense, half sparse columns: both paths inside one workgroup
    b2[1000:, :2] += 5000.0
    out = S.box_iou_rotated(cu(b1), cu(b2)).cpu().numpy()
    ref = oracle.box_iou_rotated(b1, b2, sort_mode=oracle.SORT_GPU)
    assert (bits(out) != bits(ref)).sum() == 0 and (out[:, 1000:] == 0).all()


def _degenerate_boxes(rng, n):
    """boxes whose pairs produce MORE than 8 candidate points (duplicates, quarter-turn copies, shared edges and corners):
    the dense IoU pass gives a lane 8 point slots and hands such pairs to the 24-slot redo path"""
    base = rand_rboxes(rng, n // 4, span=60, lo=8, hi=30)
    base[:, 4] = rng.choice(np.array([0.0, np.pi / 2, np.pi / 4, 0.3], np.float32), n // 4)
    dup = base.copy()
    turn = base.copy(); turn[:, 4] += np.float32(np.pi / 2); turn[:, [2, 3]] = turn[:, [3, 2]]      # same footprint
    touch = base.copy(); touch[:, 0] += touch[:, 2] * np.cos(touch[:, 4]); touch[:, 1] += touch[:, 2] * np.sin(touch[:, 4])
    return np.concatenate([base, dup, turn, touch]).astype(np.float32)


@pytest.mark.parametrize("path", ["forked", "unforked", "cols"])
def test_iou_pairs_with_many_candidate_points(rng, monkeypatch, path):
    """every way box_iou_rotated is evaluated: forked (paced zero-fill beside the readlane pair finder; the default from
    8 MB of output), un-forked (the column-major cull stores the zeros itself; small outputs), and the column-major cull
    beside the fill (A/B switch)"""
    import s2anet_amd as S
    monkeypatch.setenv("S2A_IOU_FORK", "0" if path == "unforked" else "1")
    monkeypatch.setenv("S2A_IOU_CULL_COLS", "1" if path == "cols" else "0")
    b1 = _degenerate_boxes(rng, 1600)
    b2 = np.concatenate([b1[::2], rand_rboxes(rng, 803, span=60, lo=8, hi=30)])     # 1603 columns: unaligned rows
    out = S.box_iou_rotated(cu(b1), cu(b2)).cpu().numpy()
    ref = oracle.box_iou_rotated(b1, b2, sort_mode=oracle.SORT_GPU)
    assert (bits(out) != bits(ref)).sum() == 0
    assert (np.abs(out - 1.0) < 1e-5).sum() >= 400                      # the duplicates are there
    out4 = S.box_iou_rotated(cu(b1), cu(b2[:1600])).cpu().numpy()       # 16-byte aligned rows
    assert (bits(out4) != bits(ref[:, :1600])).sum() == 0


@pytest.mark.parametrize("path", ["forked", "unforked", "cols"])
def test_iou_dense_overlap_both_paths(rng, monkeypatch, path):
    import s2anet_amd as S
    monkeypatch.setenv("S2A_IOU_FORK", "0" if path == "unforked" else "1")
    monkeypatch.setenv("S2A_IOU_CULL_COLS", "1" if path == "cols" else "0")
    b1, b2 = rand_rboxes(rng, 900, span=25, lo=30, hi=60), rand_rboxes(rng, 2300, span=25, lo=30, hi=60)
    b2[1500:, :2] += 5000.0
    out = S.box_iou_rotated(cu(b1), cu(b2)).cpu().numpy()
    ref = oracle.box_iou_rotated(b1, b2, sort_mode=oracle.SORT_GPU)
    assert (bits(out) != bits(ref)).sum() == 0 and (out[:, 1500:] == 0).all() and (out[:, :1500] > 0).mean() > 0.95


@pytest.mark.parametrize("lanes", ["0", "1"])
def test_nms_both_cull_forms_vs_oracle(rng, monkeypatch, lanes):
    """the four-waves-per-tile cull (small inputs, blocks in score order) and the one-wave-per-tile cull of large inputs
    (spatial order, columns rotating through the lanes, area-ratio test in the first stage), each forced on the same
    clustered input with ragged segments (diagonal tiles, partly filled last blocks, invalid columns)"""
    from s2anet_amd.rotated import ml_nms_rotated
    monkeypatch.setenv("S2A_NMS_CULL_LANES", "0" if lanes == "0" else "1")
    monkeypatch.setenv("S2A_NMS_SPATIAL", "1" if lanes != "0" else "0")
    n = 7001
    d = rand_rboxes(rng, n, span=300, lo=6, hi=60)
    d[:1500, :2] = d[:1500, :2] * 0.1 + 40.0            # a dense cluster: many survivors per tile
    sc = distinct_scores(rng, n)
    lab = rng.integers(0, 5, n).astype(np.float32)
    lab[:40] = 7.0                                      # a segment smaller than one block
    for thr in (0.3, 0.7):
        keep = ml_nms_rotated(cu(d), cu(sc), cu(lab), thr).cpu().numpy()
        ref = oracle.nms_rotated(d, sc, thr, labels=lab, rule=oracle.RULE_GT, sort_mode=oracle.SORT_GPU)
        assert np.array_equal(keep, ref), (lanes, thr)


def test_nms_with_duplicates_and_shared_edges(rng):
    from s2anet_amd.rotated import ml_nms_rotated
    d = _degenerate_boxes(rng, 4000)
    sc = distinct_scores(rng, len(d))
    lab = rng.integers(0, 3, len(d)).astype(np.float32)
    for thr in (0.1, 0.5, 0.9):
        keep = ml_nms_rotated(cu(d), cu(sc), cu(lab), thr).cpu().numpy()
        ref = oracle.nms_rotated(d, sc, thr, labels=lab, rule=oracle.RULE_GT, sort_mode=oracle.SORT_GPU)
        assert np.array_equal(keep, ref), thr


# ------------------------------------------------------------------ NMS
def test_iou_exotic_inputs_all_pair_finders(rng, monkeypatch):
    """large outputs (>= 8 MB: the forked path with the readlane pair finder) against the un-forked pipeline and the
    column-major finder beside the fill, and against the oracle, with (a) one huge column, (b) non-finite columns and rows
    (evaluated, never culled: NaN in, NaN out), (c) all centres on one point, (d) a far outlier"""
    import s2anet_amd as S
    n, m = 1500, 1600
    for case in ("huge", "nonfinite", "point", "outlier"):
        b1, b2 = rand_rboxes(rng, n, span=600), rand_rboxes(rng, m, span=600)
        if case == "huge":
            b2[7, 2:4] = (3000.0, 2500.0)
        elif case == "nonfinite":
            b2[3, 0], b2[11, 2], b2[500, 4], b2[900, 1] = np.nan, np.nan, np.nan, np.inf
            b1[5, 1], b1[77, 3] = np.nan, np.inf
        elif case == "point":
            b1[:, :2], b2[:, :2] = 300.0, 300.0
        else:
            b2[0, :2] = 4e6
        ref = oracle.box_iou_rotated(b1, b2, sort_mode=oracle.SORT_GPU)
        got = S.box_iou_rotated(cu(b1), cu(b2)).cpu().numpy()
        for env in ({"S2A_IOU_FORK": "0"}, {"S2A_IOU_CULL_COLS": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            other = S.box_iou_rotated(cu(b1), cu(b2)).cpu().numpy()
            for k in env:
                monkeypatch.delenv(k)
            # the pair finders must agree bit for bit, NaN patterns included (same evaluation of whatever survives)
            assert np.array_equal(got.view(np.uint32), other.view(np.uint32)), (case, env)
        # and every pair of FINITE boxes equals the oracle (what a non-finite box yields -- NaN or 0 -- is pinned by
        # nothing in the reference: its CUDA op was never run on such input; here they are evaluated, never culled)
        fin = np.isfinite(b1).all(1)[:, None] & np.isfinite(b2).all(1)[None, :]
        assert np.array_equal(got[fin].view(np.uint32), ref[fin].view(np.uint32)), case
        assert (ref[fin] > 0).mean() > (0.5 if case == "point" else 0.005)


@pytest.mark.parametrize("thr", [0.1, 0.5])
def test_nms_golden_keep_bitexact(thr):
    import s2anet_amd as S
    from s2anet_amd.rotated import nms_rotated_raw
    g = golden("nms_2k.npz")
    d, s, lab = cu(g["dets"]), cu(g["scores"]), cu(g["labels"])
    k = S.ml_nms_rotated(d, s, lab, thr).cpu().numpy()
    assert k.dtype == np.int64 and np.array_equal(k, g[f"ml_keep_gt_{thr}"])
    k1 = nms_rotated_raw(d, s, thr).cpu().numpy()
    assert np.array_equal(k1, g[f"sc_keep_gt_{thr}"])
    # the reference CPU op (>= rule, std::sort branch) keeps the same boxes on this fixture
    assert np.array_equal(k, g[f"ml_keep_ge_{thr}"])
    assert np.array_equal(k1, g[f"sc_keep_ge_{thr}"])


@pytest.mark.parametrize("thr", [0.03, 0.06, 0.1, 0.3, 0.5, 0.7])
def test_nms_area_ratio_shortcut_is_exact(rng, thr):
    """the cull kernel drops pairs whose area ratio cannot reach the threshold (IoU <= min / max) before the dense IoU
    pass: keep lists must stay those of the oracle for every threshold, with widely spread sizes, concentric boxes
    whose ratio sits right at the threshold, and very thin boxes (for which the shortcut is switched off)"""
    import s2anet_amd as S
    n = 2500
    d = rand_rboxes(rng, n, span=300, lo=2, hi=120)
    d[: n // 4, 2] *= rng.uniform(0.01, 0.05, n // 4).astype(np.float32)            # very thin
    # concentric pairs with area ratio thr * (1 +- small): the shortcut's margin must not flip them
    m = 300
    base = rand_rboxes(rng, m, span=300, lo=30, hi=80)
    inner = base.copy()
    f = np.sqrt(thr * (1 + rng.uniform(-0.03, 0.03, m))).astype(np.float32)
    inner[:, 2] *= f
    inner[:, 3] *= f
    d = np.concatenate([d, base, inner]).astype(np.float32)
    s = distinct_scores(rng, d.shape[0])
    lab = rng.integers(0, 3, d.shape[0]).astype(np.float32)
    lab[n:n + m] = lab[n + m:]                                                     # the pairs share a label
    k = S.ml_nms_rotated(cu(d), cu(s), cu(lab), float(thr)).cpu().numpy()
    ref = oracle.ml_nms_rotated(d, s, lab, float(thr), rule=oracle.RULE_GT, sort_mode=oracle.SORT_GPU, cull=True)
    assert np.array_equal(k, ref)


def test_nms_long_chain_needs_more_rounds_than_launched(rng):
    """a chain of boxes in which every box overlaps only its successor above the threshold (shift 0.3 w: IoU 0.54 with
    the next, 0.25 with the one after) and scores fall along the chain: the greedy result alternates keep / drop and the
    decision of box k depends on box k - 1 -- hundreds of dependent rounds, far more than the launched ones, so the
    single-workgroup clean-up kernel finishes it.  Several chains with different labels, some reversed (scores RISE
    along the chain: everything but the ends of the overlaps is decided in the first round), plus random clutter."""
    import s2anet_amd as S
    boxes, scores, labels = [], [], []
    for c, (length, rev) in enumerate([(700, False), (333, True), (64, False), (65, False), (2, False), (1500, False)]):
        k = np.arange(length, dtype=np.float32)
        b = np.stack([100 + 0.3 * 40 * k, np.full(length, 50.0 + 200 * c, np.float32), np.full(length, 40.0, np.float32),
                      np.full(length, 20.0, np.float32), np.zeros(length, np.float32)], 1).astype(np.float32)
        sc = np.linspace(0.9, 0.1, length).astype(np.float32) + np.float32(c) * np.float32(1e-4)
        boxes.append(b), scores.append(sc[::-1] if rev else sc), labels.append(np.full(length, c % 3, np.float32))
    clutter = rand_rboxes(rng, 2000, span=3000)
    boxes.append(clutter), scores.append(rng.uniform(0.05, 0.95, 2000).astype(np.float32)), labels.append(rng.integers(0, 3, 2000).astype(np.float32))
    d, s, lab = np.concatenate(boxes), np.concatenate(scores), np.concatenate(labels)
    s = (s + np.arange(len(s), dtype=np.float32) * np.float32(1e-7)).astype(np.float32)      # distinct
    assert len(np.unique(s)) == len(s)
    keep = S.ml_nms_rotated(cu(d), cu(s), cu(lab), 0.5).cpu().numpy()
    ref = oracle.ml_nms_rotated(d, s, lab, 0.5)
    assert np.array_equal(keep, ref)
    first = np.arange(700)
    assert np.isin(first[::2], keep).all() and not np.isin(first[1::2], keep).any()          # the alternating pattern


def test_nms_list_overflow_takes_the_direct_path(rng):
    """a workspace sized for sparse segments (s2a_nms_rotated_workspace_bytes(n, 1)) on an input in which nearly every
    pair overlaps: pair and edge lists overflow, the direct greedy kernel redoes the segments -- same keep flags as the
    oracle, no error, nothing silently dropped"""
    import ctypes
    from s2anet_amd import _lib
    L = _lib.lib()
    n, nseg = 2500, 3
    d = rand_rboxes(rng, n, span=60, lo=30, hi=80)                                           # everything piles up
    s = distinct_scores(rng, n)
    seg = rng.integers(0, nseg, n).astype(np.int32)
    seg[rng.integers(0, n, 50)] = -1                                                         # some ignored rows
    D, Sc, Sg = cu(d), cu(s), cu(seg)
    flags = torch.zeros(n, dtype=torch.uint8, device=dev())
    small = L.s2a_nms_rotated_workspace_bytes(n, 1)
    big = L.s2a_nms_rotated_workspace_bytes(n, n)
    assert small < big
    out = {}
    for name, nbytes in (("small", small), ("big", big)):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
        _lib.check(L.s2a_nms_rotated_segmented(_lib.ptr(D), _lib.ptr(Sc), _lib.ptr(Sg), None, n, nseg, 1, 0.3,
                                               _lib.ptr(flags), None, None, 0, _lib.ptr(ws), ws.numel(),
                                               _lib.stream_ptr(dev())))
        out[name] = flags.cpu().numpy().astype(bool)
    ref = np.zeros(n, bool)
    for c in range(nseg):
        idx = np.nonzero(seg == c)[0]
        ref[idx[oracle.nms_rotated(d[idx], s[idx], 0.3)]] = True
    assert np.array_equal(out["big"], ref) and np.array_equal(out["small"], ref)
    assert ref.sum() < n // 10 and not ref[seg < 0].any()


def _small_stats():
    import ctypes
    from s2anet_amd import _lib
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(_lib.lib().s2a_nms_small_stats(ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def test_nms_small_inputs_one_launch(rng, monkeypatch):
    """round 5: synchronous nms_rotated / ml_nms_rotated calls on <= 16 384 rows are settled by ONE kernel (k_nms_small: a
    workgroup per label -- LDS sort, cull, exact IoU into an LDS bit mask, greedy walk, last-workgroup merge) when they fit
    its limits, and by the general path otherwise.  Keep lists == oracle == general path (S2A_NMS_SMALL=0) for: typical
    detector-sized inputs, one row, ties, duplicates / shared edges (> 8 candidate points: the 24-slot redo), piles (candidate
    list overflow -> fallback), a label with > 640 rows (fallback), 65 and 200 distinct labels (fallback), weird labels."""
    import s2anet_amd as S
    from s2anet_amd.rotated import nms_rotated_raw

    def both(d, s, lab, thr):
        t0, f0 = _small_stats()
        got = (S.ml_nms_rotated(cu(d), cu(s), cu(lab), thr) if lab is not None else nms_rotated_raw(cu(d), cu(s), thr)).cpu().numpy()
        t1, f1 = _small_stats()
        monkeypatch.setenv("S2A_NMS_SMALL", "0")
        ref = (S.ml_nms_rotated(cu(d), cu(s), cu(lab), thr) if lab is not None else nms_rotated_raw(cu(d), cu(s), thr)).cpu().numpy()
        monkeypatch.delenv("S2A_NMS_SMALL")
        t2, f2 = _small_stats()
        assert (t2, f2) == (t1, f1)                                  # the switch really bypasses the small path
        want = oracle.ml_nms_rotated(d, s, lab, thr) if lab is not None else oracle.nms_rotated(d, s, thr)
        assert np.array_equal(got, want) and np.array_equal(ref, want)
        return t1 - t0, f1 - f0

    # the size one image's post-processing has: 5 000 rows, 15 labels -> one launch
    n = 5000
    d, s = rand_rboxes(rng, n), distinct_scores(rng, n)
    lab = rng.integers(0, 15, n).astype(np.float32)
    assert both(d, s, lab, 0.5) == (1, 0)
    assert both(d, s, lab, 0.1) == (1, 0)
    # 16 000 rows over 40 labels (400 each), negative / fractional / huge labels, -0.0 and +0.0 one label
    n = 16000
    d, s = rand_rboxes(rng, n, span=2000), distinct_scores(rng, n)
    vals = np.concatenate([np.arange(36, dtype=np.float32) * 0.5 - 3.0, np.array([1e9, -1e-30, 65504.0, 3.0000002], np.float32)])
    lab = vals[rng.integers(0, 40, n)]
    lab[lab == 0.0] = np.where(rng.random((lab == 0.0).sum()) < 0.5, np.float32(-0.0), np.float32(0.0))
    assert both(d, s, lab, 0.3) == (1, 0)
    # single class, no labels: one segment of <= 640 rows; one row; score ties (broken by the original row)
    for m in (1, 2, 63, 64, 65, 640):
        d, s = rand_rboxes(rng, m, span=120 * max(1.0, (m / 16) ** 0.5)), distinct_scores(rng, m)
        if m >= 63:
            s[5] = s[17] = s[40]
        assert both(d, s, None, 0.3) == (1, 0)
    # duplicates, quarter-turn copies, boxes sharing an edge: more than 8 candidate points -> the 24-slot redo inside the kernel
    base = rand_rboxes(rng, 150, span=300, lo=20, hi=60)
    dup = base.copy()
    turn = base.copy(); turn[:, 4] += np.float32(np.pi / 2)
    edge = base.copy(); edge[:, 0] += base[:, 2] * np.cos(base[:, 4]); edge[:, 1] += base[:, 2] * np.sin(base[:, 4])
    d = np.concatenate([base, dup, turn, edge]).astype(np.float32)
    s = distinct_scores(rng, len(d))
    assert both(d, s, (np.arange(len(d)) % 2).astype(np.float32), 0.5) == (1, 0)
    # everything piles up: more than 4 096 surviving pairs in a segment -> status -> general path, same answer
    d, s = rand_rboxes(rng, 600, span=50, lo=30, hi=80), distinct_scores(rng, 600)
    assert both(d, s, np.zeros(600, np.float32), 0.3) == (0, 1)
    # a label with more than 640 rows; 65 labels; 200 labels
    n = 3000
    d, s = rand_rboxes(rng, n), distinct_scores(rng, n)
    assert both(d, s, (np.arange(n) < 700).astype(np.float32), 0.5) == (0, 1)
    assert both(d, s, (np.arange(n) % 65).astype(np.float32), 0.5) == (0, 1)
    assert both(d, s, (np.arange(n) % 200).astype(np.float32), 0.5) == (0, 1)
    # beyond 16 384 rows the general path is taken without trying
    n = 17000
    d, s = rand_rboxes(rng, n, span=2000), distinct_scores(rng, n)
    assert both(d, s, rng.integers(0, 40, n).astype(np.float32), 0.5) == (0, 0)


def _segmented_dets(L, _lib, D, Sc, Sg, Gr, Cl, n, nseg, ngrp, thr, K):
    wire = torch.empty((ngrp, K * 7 + 1), dtype=torch.float32, device=dev())
    labels = torch.empty((ngrp, K), dtype=torch.int32, device=dev())
    counts = torch.empty((ngrp,), dtype=torch.int32, device=dev())
    ws = torch.empty(L.s2a_nms_rotated_workspace_bytes(n, n), dtype=torch.uint8, device=dev())
    _lib.check(L.s2a_nms_rotated_segmented_dets(_lib.ptr(D), _lib.ptr(Sc), _lib.ptr(Sg), _lib.ptr(Gr), _lib.ptr(Cl), n, nseg, ngrp,
                                                thr, K, _lib.ptr(wire), _lib.ptr(labels), _lib.ptr(counts), None, None, None,
                                                _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev())))
    torch.cuda.synchronize()
    return wire.cpu().numpy(), labels.cpu().numpy(), counts.cpu().numpy()


@pytest.mark.parametrize("case", ["detector_like", "huge_segment", "many_kept", "tiny"])
def test_nms_own_segment_sort_and_scanning_emit(rng, monkeypatch, case):
    """round 5: the (segment, score) order by k_seg_hist / k_seg_sort (one workgroup per segment, LDS bitonic sort; a segment
    beyond the LDS capacity ranks its rows by counting) and the detection rows by k_nms_group_emit_scan (kept rows of a group
    collected and sorted in LDS, folded in rounds when more are kept than the LDS holds) -- against the library-sort form
    (S2A_NMS_SEGSORT=0) BIT FOR BIT and against the oracle per segment.  Padding rows anywhere, empty segments, segments of
    one row, a segment of 9 000 rows, 11 000 kept rows in one group."""
    from s2anet_amd import _lib
    L = _lib.lib()
    if case == "detector_like":        # 3 images x 15 classes inside a mostly empty static buffer, padding at the end
        n_real, n, nseg, ngrp, K, span = 9000, 40000, 45, 3, 700, 400.0
    elif case == "huge_segment":       # one segment beyond kSegCap = 8192 rows (rank-by-counting path), the rest small
        n_real, n, nseg, ngrp, K, span = 12000, 12500, 6, 2, 2000, 900.0
    elif case == "many_kept":          # nothing overlaps: > 8192 kept rows in one group -> the emit folds in rounds
        n_real, n, nseg, ngrp, K, span = 11000, 11100, 4, 1, 300, 20000.0
    else:
        n_real, n, nseg, ngrp, K, span = 37, 64, 9, 2, 5, 80.0
    d = np.zeros((n, 5), np.float32)
    d[:n_real] = rand_rboxes(rng, n_real, span=span, lo=4, hi=(12 if case == "many_kept" else 80))
    s = np.full(n, -1.0, np.float32)
    s[:n_real] = distinct_scores(rng, n_real)
    if case == "tiny":
        s[3] = s[4] = s[5]                                                  # score ties: broken by the original row
    seg = np.full(n, -1, np.int32)
    if case == "huge_segment":
        seg[:n_real] = np.where(np.arange(n_real) < 9000, 2, rng.integers(0, nseg, n_real))
    else:
        seg[:n_real] = rng.integers(0, nseg, n_real)
        seg[:n_real][seg[:n_real] == 1] = 0                                 # segment 1 stays EMPTY
    per = (nseg + ngrp - 1) // ngrp
    grp = np.where(seg >= 0, seg // per, -1).astype(np.int32)
    cls = np.where(seg >= 0, seg % per, -1).astype(np.int32)
    hole = rng.integers(0, n_real, max(n_real // 50, 2))                    # padding rows in the middle of the buffer as well
    seg[hole] = grp[hole] = cls[hole] = -1
    order = rng.permutation(n) if case != "detector_like" else np.arange(n)
    d, s, seg, grp, cls = d[order], s[order], seg[order], grp[order], cls[order]
    D, Sc, Sg, Gr, Cl = cu(d), cu(s), cu(seg), cu(grp), cu(cls)
    thr = 0.3
    monkeypatch.setenv("S2A_NMS_SEGSORT", "0")
    w0, l0, c0 = _segmented_dets(L, _lib, D, Sc, Sg, Gr, Cl, n, nseg, ngrp, thr, K)
    monkeypatch.delenv("S2A_NMS_SEGSORT")
    w1, l1, c1 = _segmented_dets(L, _lib, D, Sc, Sg, Gr, Cl, n, nseg, ngrp, thr, K)
    assert np.array_equal(c0, c1) and np.array_equal(l0, l1) and np.array_equal(w0.view(np.uint32), w1.view(np.uint32))
    # and the oracle: per segment NMS, per group the kept rows by descending score (ties: original row), cut to K
    keep = np.zeros(n, bool)
    for c in range(nseg):
        idx = np.nonzero(seg == c)[0]
        if len(idx):
            keep[idx[oracle.nms_rotated(d[idx], s[idx], thr)]] = True
    for g in range(ngrp):
        idx = np.nonzero(keep & (grp == g))[0]
        idx = idx[np.lexsort((idx, -s[idx].astype(np.float64)))][:K]
        assert c1[g] == len(idx)
        rows = w1[g, :K * 7].reshape(K, 7)
        assert np.array_equal(rows[:len(idx), :5], d[idx]) and np.array_equal(rows[:len(idx), 5], s[idx])
        assert np.array_equal(rows[:len(idx), 6], cls[idx].astype(np.float32)) and (rows[len(idx):, 6] == -1).all()
    if case == "many_kept":
        assert keep.sum() > 8192
    if case == "huge_segment":
        assert (seg == 2).sum() > 8192


def test_nms_segmented_big_segments_take_the_spatial_path(rng, monkeypatch):
    """the segmented entry point with few, large segments (> 4096 rows on average: Morton order + bounding-box tile
    filter) and with many small ones (blocks in score order, every tile tested) gives the oracle's keep flags either way"""
    from s2anet_amd import _lib
    L = _lib.lib()
    for n, nseg, span in ((30000, 3, 700.0), (30000, 40, 250.0)):
        d = rand_rboxes(rng, n, span=span)
        s = distinct_scores(rng, n)
        seg = rng.integers(0, nseg, n).astype(np.int32)
        ws = torch.empty(L.s2a_nms_rotated_workspace_bytes(n, n), dtype=torch.uint8, device=dev())
        D, Sc, Sg = cu(d), cu(s), cu(seg)                     # (held: the raw pointers below do not keep them alive)
        ref = np.zeros(n, bool)
        for c in range(nseg):
            idx = np.nonzero(seg == c)[0]
            ref[idx[oracle.nms_rotated(d[idx], s[idx], 0.5)]] = True
        assert n // 4 < ref.sum() < n
        for fork in ("0", "1"):          # one stream (the segmented call's default) and the side-stream form of the drop-in ops
            monkeypatch.setenv("S2A_NMS_FORK", fork)
            flags = torch.zeros(n, dtype=torch.uint8, device=dev())
            _lib.check(L.s2a_nms_rotated_segmented(_lib.ptr(D), _lib.ptr(Sc), _lib.ptr(Sg), None, n, nseg, 1, 0.5,
                                                   _lib.ptr(flags), None, None, 0, _lib.ptr(ws), ws.numel(),
                                                   _lib.stream_ptr(dev())))
            got = flags.cpu().numpy().astype(bool)
            assert np.array_equal(got, ref), (n, nseg, fork, int((got != ref).sum()))


@pytest.mark.parametrize("thr", [0.1, 0.3, 0.5, 0.75])
def test_nms_iou_upper_bound_prefilter_is_exact(rng, thr):
    """pairs whose IoU sits right at the threshold, where the projected-overlap bound is tight (same-size boxes shifted
    along an edge, slightly rotated, nested boxes, crosses): the pre-filter (rbox_geom.hpp:nms_pair_skippable) may only
    drop pairs that the full evaluation would not have suppressed -> keep lists identical to the oracle"""
    import s2anet_amd as S
    m = 1500
    w, h = rng.uniform(10, 60, m).astype(np.float32), rng.uniform(10, 60, m).astype(np.float32)
    cx, cy = (np.arange(m) % 40 * 200.0).astype(np.float32), (np.arange(m) // 40 * 200.0).astype(np.float32)
    ang = rng.uniform(-0.7, 2.3, m).astype(np.float32)
    a = np.stack([cx, cy, w, h, ang], 1)
    # partner: shifted along the box's own w axis so that the IoU of two equal boxes, (w - t) / (w + t), lands within
    # +-2 % of thr; a third of them also shrunk (nested: IoU = area ratio) or turned by a small angle
    eps = rng.uniform(-0.02, 0.02, m).astype(np.float32)
    t = w * (1 - thr * (1 + eps)) / (1 + thr * (1 + eps))
    b = a.copy()
    b[:, 0] += t * np.cos(ang)
    b[:, 1] += t * np.sin(ang)
    kind = np.arange(m) % 3
    nest = kind == 1
    b[nest, :2] = a[nest, :2]
    b[nest, 2] = a[nest, 2] * np.sqrt(thr * (1 + eps[nest]))
    b[nest, 3] = a[nest, 3] * np.sqrt(thr * (1 + eps[nest]))
    b[kind == 2, 4] += rng.uniform(-0.05, 0.05, int((kind == 2).sum())).astype(np.float32)
    d = np.concatenate([a, b]).astype(np.float32)
    s = distinct_scores(rng, 2 * m)
    lab = np.zeros(2 * m, np.float32)
    keep = S.ml_nms_rotated(cu(d), cu(s), cu(lab), thr).cpu().numpy()
    ref = oracle.ml_nms_rotated(d, s, lab, thr)
    assert np.array_equal(keep, ref)
    assert m + m // 4 < len(ref) < 2 * m - m // 4                                            # both outcomes occur often


def test_nms_spatial_filter_degenerate_geometry(rng):
    """the block bounding-box filter must never drop a pair the reference would evaluate: all centres identical
    (zero-size bounding box), a huge coordinate range with a dense cluster in one Morton cell, non-finite boxes (a NaN
    IoU never suppresses, as in the reference), one row per segment, and boxes much larger than the blocks' extent"""
    import s2anet_amd as S
    # identical centres, different sizes / angles
    n = 700
    d = rand_rboxes(rng, n, span=1.0)
    d[:, :2] = 512.0
    s, lab = distinct_scores(rng, n), rng.integers(0, 2, n).astype(np.float32)
    assert np.array_equal(S.ml_nms_rotated(cu(d), cu(s), cu(lab), 0.5).cpu().numpy(), oracle.ml_nms_rotated(d, s, lab, 0.5))
    # one far outlier stretches the quantisation grid: the cluster collapses into a single cell
    d = rand_rboxes(rng, 3000, span=300)
    d[0, :2] = 1e7
    s, lab = distinct_scores(rng, 3000), rng.integers(0, 4, 3000).astype(np.float32)
    assert np.array_equal(S.ml_nms_rotated(cu(d), cu(s), cu(lab), 0.4).cpu().numpy(), oracle.ml_nms_rotated(d, s, lab, 0.4))
    # non-finite rows: kept (nothing can suppress them, they suppress nothing)
    d = rand_rboxes(rng, 500, span=100)
    d[5, 0], d[17, 2], d[40, 4], d[77, 1] = np.nan, np.nan, np.nan, np.inf
    s, lab = distinct_scores(rng, 500), np.zeros(500, np.float32)
    keep = S.ml_nms_rotated(cu(d), cu(s), cu(lab), 0.3).cpu().numpy()
    ok = np.ones(500, bool)
    ok[[5, 17, 40, 77]] = False
    ref = oracle.ml_nms_rotated(d[ok], s[ok], lab[ok], 0.3)
    assert np.isin([5, 17, 40, 77], keep).all()
    assert np.array_equal(keep[~np.isin(keep, [5, 17, 40, 77])], np.nonzero(ok)[0][ref])
    # every row its own label: nothing is ever compared
    d = rand_rboxes(rng, 300, span=20)
    s, lab = distinct_scores(rng, 300), np.arange(300, dtype=np.float32)
    keep = S.ml_nms_rotated(cu(d), cu(s), cu(lab), 0.1).cpu().numpy()
    assert np.array_equal(keep, np.argsort(-s, kind="stable"))
    # boxes far larger than a block's extent, single class (nms_rotated)
    d = rand_rboxes(rng, 2000, span=2000, lo=200, hi=1500)
    s = distinct_scores(rng, 2000)
    from s2anet_amd.rotated import nms_rotated_raw
    assert np.array_equal(nms_rotated_raw(cu(d), cu(s), 0.6).cpu().numpy(), oracle.nms_rotated(d, s, 0.6))


def test_nms_four_box_case_and_wrapper_quirks():
    import s2anet_amd as S
    from s2anet_amd.rotated import nms_rotated_raw
    g = golden("nms_2k.npz")
    d, s, lab = cu(g["d4"]), cu(g["s4"]), cu(g["l4"])
    assert S.ml_nms_rotated(d, s, lab, 0.5).tolist() == [0, 2, 3]
    assert nms_rotated_raw(d, s, 0.5).tolist() == [0, 3]
    assert nms_rotated_raw(d, s, 1.0).tolist() == [0, 1, 2, 3]      # strict > (GPU rule)
    dets6 = torch.cat([d, s[:, None]], 1)
    kept, inds = S.nms_rotated(dets6, 0.5)
    assert inds.tolist() == [0, 3] and kept.shape == (2, 6)
    empty = torch.zeros((0, 6), device=dev())
    assert S.nms_rotated(empty, 0.5) is empty
    assert S.ml_nms_rotated(empty[:, :5], empty[:, 5], empty[:, 5], 0.5).shape == (0,)
    # f16 scores are only sorted
    assert S.ml_nms_rotated(d, s.half(), lab, 0.5).tolist() == [0, 2, 3]


@pytest.mark.parametrize("n,span,nl", [(6000, 500, 15), (3000, 120, 3), (257, 40, 1), (64, 30, 2), (1, 10, 1)])
def test_nms_random_vs_oracle(rng, n, span, nl):
    import s2anet_amd as S
    from s2anet_amd.rotated import nms_rotated_raw
    d, s = rand_rboxes(rng, n, span=span), distinct_scores(rng, n)
    lab = rng.integers(0, nl, n).astype(np.float32)
    k = S.ml_nms_rotated(cu(d), cu(s), cu(lab), 0.5).cpu().numpy()
    ref = oracle.ml_nms_rotated(d, s, lab, 0.5, rule=oracle.RULE_GT, sort_mode=oracle.SORT_GPU, cull=True)
    assert np.array_equal(k, ref), (len(k), len(ref))
    k1 = nms_rotated_raw(cu(d), cu(s), 0.3).cpu().numpy()
    ref1 = oracle.nms_rotated(d, s, 0.3, rule=oracle.RULE_GT, sort_mode=oracle.SORT_GPU, cull=True)
    assert np.array_equal(k1, ref1)


def test_nms_properties_at_scale(rng):
    """size-independent properties at a size the oracle cannot scan pair by pair quickly:
    idempotence (NMS of the kept set keeps everything), order, label independence."""
    import s2anet_amd as S
    n = 60000
    d, s = rand_rboxes(rng, n, span=1024), distinct_scores(rng, n)
    lab = rng.integers(0, 15, n).astype(np.float32)
    D, Sc, Lb = cu(d), cu(s), cu(lab)
    k = S.ml_nms_rotated(D, Sc, Lb, 0.5)
    ks = Sc[k]
    assert (ks[1:] < ks[:-1]).all()                      # descending score order
    k2 = S.ml_nms_rotated(D[k], Sc[k], Lb[k], 0.5)
    assert k2.numel() == k.numel() and (k2 == torch.arange(k.numel(), device=k.device)).all()
    # per-label decomposition: ml-NMS == union of single-class NMS per label
    from s2anet_amd.rotated import nms_rotated_raw
    parts = []
    for c in range(15):
        idx = (Lb == c).nonzero()[:, 0]
        parts.append(idx[nms_rotated_raw(D[idx], Sc[idx], 0.5)])
    allk = torch.cat(parts)
    assert torch.equal(torch.sort(allk)[0], torch.sort(k)[0])
    ref = oracle.ml_nms_rotated(d, s, lab, 0.5, rule=oracle.RULE_GT, sort_mode=oracle.SORT_GPU, cull=True)
    assert np.array_equal(k.cpu().numpy(), ref)


def test_multiclass_nms_matches_reference_python():
    import s2anet_amd as S
    g = golden("head_glue.npz")
    det, lab = S.multiclass_nms_rotated(cu(g["mc_bboxes"]), cu(g["mc_scores"]), 0.05, 0.5, 300)
    assert np.array_equal(det.cpu().numpy(), g["mc_det"])
    assert np.array_equal(lab.cpu().numpy(), g["mc_labels"])
    det0, lab0 = S.multiclass_nms_rotated(cu(g["mc_bboxes"]), cu(g["mc_scores"]) * 0, 0.05, 0.5, 300)
    assert tuple(det0.shape) == tuple(g["mc_empty_det_shape"]) and tuple(lab0.shape) == tuple(g["mc_empty_lab_shape"])


def test_batched_multiclass_nms_equals_per_image(rng):
    import s2anet_amd as S
    B, n, C = 3, 700, 15
    boxes = np.stack([rand_rboxes(rng, n, span=260) for _ in range(B)])
    scores = (rng.random((B, n, C)) ** 8).astype(np.float32)
    scores[2] *= 0.01                                    # an image without candidates
    for cap in (None, 8000):      # 8000 >= all candidates of the batch: lossless
        dets, labels, counts = S.batched_multiclass_nms_rotated(cu(boxes), cu(scores), 0.05, 0.5, 200,
                                                                max_candidates=cap)
        for b in range(B):
            rd, rl = oracle.multiclass_nms_rotated(boxes[b], scores[b], 0.05, 0.5, 200)
            kb = int(counts[b])
            assert kb == rd.shape[0]
            assert np.array_equal(dets[b, :kb].cpu().numpy(), rd)
            assert np.array_equal(labels[b, :kb].cpu().numpy().astype(np.float32), rl)
            assert (labels[b, kb:] == -1).all()


def test_batched_nms_uncapped_call_at_benchmark_shape_takes_the_cell_ordered_path(rng):
    """max_candidates=None at 8 x 5344 x 15 sizes the candidate buffer at 641 280 rows, ~500 k of them padding (segment
    -1): the call goes down the cell-ordered path. Results equal the oracle per image, and the padding rows do not queue
    for one histogram counter (that cost 6.7 ms once; the bound here is generous, a regression is 10x over it)"""
    import s2anet_amd as S, time
    B, n, C = 8, 5344, 15
    boxes = np.stack([np.concatenate([rng.uniform(0, 1024, (n, 2)), rng.uniform(8, 80, (n, 2)),
                                      rng.uniform(-0.7, 2.3, (n, 1))], 1) for _ in range(B)]).astype(np.float32)
    scores = (rng.random((B, n, C)) ** 12).astype(np.float32)
    bb, sc = cu(boxes), cu(scores)
    for _ in range(3):
        out = S.batched_multiclass_nms_rotated(bb, sc, 0.05, 0.5, 2000)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        out = S.batched_multiclass_nms_rotated(bb, sc, 0.05, 0.5, 2000)
    torch.cuda.synchronize()
    assert (time.perf_counter() - t) / 5 < 3e-3
    dets, labels, counts = out
    for b in (0, 3, 7):
        rd, rl = oracle.multiclass_nms_rotated(boxes[b], scores[b], 0.05, 0.5, 2000)
        kb = int(counts[b])
        assert kb == len(rd) and np.array_equal(dets[b, :kb].cpu().numpy(), rd)
        assert np.array_equal(labels[b, :kb].cpu().numpy().astype(np.float32), rl)


def test_nms_many_rows_at_one_centre_share_a_histogram_counter(rng):
    """degenerate input for the counting sort: 40 000 boxes of one label at the same centre (one cell). The rows of a
    wave that share a cell take their ranks from one atomic; the result is the oracle's"""
    import s2anet_amd as S
    n = 40000
    d = np.zeros((n, 5), np.float32)
    d[:, 0:2] = 300.0
    d[:, 2] = rng.uniform(10, 40, n); d[:, 3] = rng.uniform(10, 40, n); d[:, 4] = rng.uniform(-0.7, 0.7, n)
    d[: n // 2, 0] += rng.uniform(0, 2000, n // 2).astype(np.float32)    # half of them spread out
    s = distinct_scores(rng, n)
    lab = np.zeros(n, np.float32)
    keep = S.ml_nms_rotated(cu(d), cu(s), cu(lab), 0.5).cpu().numpy()
    ref = oracle.ml_nms_rotated(d, s, lab, 0.5)
    assert np.array_equal(keep, ref)


def test_batched_nms_writes_the_wire_buffer_and_counts_dropped_candidates(rng):
    """the NMS finish kernel writes the detection rows once: dets / labels / counts and the all-gather wire buffer are
    the same memory or exact copies, padding rows are 0,...,0,-1, the candidate cap is accounted without a host sync"""
    import s2anet_amd as S
    from s2anet_amd.gather import unpack_detections
    B, n, C, K = 3, 500, 15, 120
    boxes = np.stack([rand_rboxes(rng, n, span=240) for _ in range(B)])
    scores = (rng.random((B, n, C)) ** 8).astype(np.float32)
    scores[1] *= 0.01                                             # an image without candidates
    found = int((scores > 0.05).sum())
    total = torch.zeros(1, dtype=torch.int64, device="cuda")
    dets, labels, counts, ovf, wire = S.batched_multiclass_nms_rotated(
        cu(boxes), cu(scores), 0.05, 0.5, K, return_overflow=True, dropped_total=total, return_wire=True)
    assert tuple(wire.shape) == (B, K * 7 + 1) and dets.data_ptr() == wire.data_ptr()      # dets is a view of the wire buffer
    d2, l2, c2 = unpack_detections(wire, K)
    assert torch.equal(d2, dets) and torch.equal(l2, labels) and torch.equal(c2, counts)
    assert ovf.cpu().tolist() == [found, 0] and int(total) == 0
    for b in range(B):
        rd, rl = oracle.multiclass_nms_rotated(boxes[b], scores[b], 0.05, 0.5, K)
        kb = int(counts[b])
        assert kb == len(rd) and np.array_equal(dets[b, :kb].cpu().numpy(), rd)
        assert np.array_equal(labels[b, :kb].cpu().numpy().astype(np.float32), rl)
        assert (labels[b, kb:] == -1).all() and (dets[b, kb:] == 0).all()
        assert (wire[b, :K * 7].view(K, 7)[kb:, 6] == -1).all()
    assert int(counts[1]) == 0
    # a cap below the candidate count: reported twice (this call, running total), never silent
    cap = found - 37
    for rep in (1, 2):
        _, _, _, ovf2 = S.batched_multiclass_nms_rotated(cu(boxes), cu(scores), 0.05, 0.5, K, max_candidates=cap,
                                                         return_overflow=True, dropped_total=total)
        assert ovf2.cpu().tolist() == [found, 37] and int(total) == 37 * rep
    # no candidate at all in the batch
    d0, l0, c0, o0 = S.batched_multiclass_nms_rotated(cu(boxes), cu(scores * 0), 0.05, 0.5, K, return_overflow=True)
    assert (d0 == 0).all() and (l0 == -1).all() and (c0 == 0).all() and o0.cpu().tolist() == [0, 0]
    # an EMPTY candidate set (no box at all: the tensors have no storage): padded empty results, zero accounting
    e_boxes = torch.empty((B, 0, 5), dtype=torch.float32, device="cuda")
    e_scores = torch.empty((B, 0, C), dtype=torch.float32, device="cuda")
    d1, l1, c1, o1, w1 = S.batched_multiclass_nms_rotated(e_boxes, e_scores, 0.05, 0.5, K, return_overflow=True,
                                                          dropped_total=total, return_wire=True)
    assert tuple(d1.shape) == (B, K, 6) and (d1 == 0).all() and (l1 == -1).all() and (c1 == 0).all()
    assert o1.cpu().tolist() == [0, 0] and int(total) == 74 and (w1[:, K * 7] == 0).all()
    assert (w1[:, :K * 7].view(B, K, 7)[..., 6] == -1).all()
    with pytest.raises(ValueError):
        S.batched_multiclass_nms_rotated(cu(boxes), cu(scores), 0.05, 0.5, K, max_candidates=0)
    # the C entry point with n = 0 and candidates that WERE found upstream (a cap of zero rows): all of them count as dropped
    from s2anet_amd import _lib
    L = _lib.lib()
    wire0 = torch.empty((B, K * 7 + 1), dtype=torch.float32, device="cuda")
    found_dev = torch.tensor([41], dtype=torch.int64, device="cuda")
    ovf_dev = torch.empty((2,), dtype=torch.int64, device="cuda")
    _lib.check(L.s2a_nms_rotated_segmented_dets(None, None, None, None, None, 0, B * C, B, 0.5, K, _lib.ptr(wire0), None, None,
                                                _lib.ptr(found_dev), _lib.ptr(ovf_dev), _lib.ptr(total), None, 0,
                                                _lib.stream_ptr(wire0.device)))
    assert ovf_dev.cpu().tolist() == [41, 41] and int(total) == 74 + 41 and (wire0[:, K * 7] == 0).all()


def test_nms_order_b_counting_sort_labels_of_any_kind(rng, monkeypatch):
    """the spatial order of big inputs is built by an own counting sort into (segment, Morton cell) buckets (k_spb_*): the
    distinct segment keys are hashed and numbered in order of arrival.  Labels that are not small integers (negative,
    fractional, huge, -0.0 next to +0.0), the rocPRIM form of the same order (S2A_NMS_SORTB=0) and -- more distinct labels
    than the table holds -- the direct fallback must all give the oracle's keep list."""
    import s2anet_amd as S
    n = 9000
    d = rand_rboxes(rng, n, span=700)
    s = distinct_scores(rng, n)
    weird = np.array([-3.5, -0.0, 0.0, 0.25, 1.0, 7.0, 1e9, -1e-30, 65504.0, 3.0000002], np.float32)
    lab = weird[rng.integers(0, len(weird), n)]
    want = oracle.ml_nms_rotated(d, s, lab, 0.5)
    for mode in ("1", "0"):
        monkeypatch.setenv("S2A_NMS_SORTB", mode)
        got = S.ml_nms_rotated(cu(d), cu(s), cu(lab), 0.5).cpu().numpy()
        assert np.array_equal(got, want), mode
    monkeypatch.delenv("S2A_NMS_SORTB")
    # 12 000 distinct labels on 24 000 rows: the 16 384-slot table takes them (load 0.73) or reports itself full; either
    # way the keep list is the oracle's
    n = 24000
    d = rand_rboxes(rng, n, span=300)
    s = distinct_scores(rng, n)
    lab = (rng.integers(0, 12000, n)).astype(np.float32) * 0.5
    got = S.ml_nms_rotated(cu(d), cu(s), cu(lab), 0.3).cpu().numpy()
    assert np.array_equal(got, oracle.ml_nms_rotated(d, s, lab, 0.3))
    # and one label for everything (plain nms_rotated): a single segment of 30 000 rows
    n = 30000
    d = rand_rboxes(rng, n, span=2000)
    s = distinct_scores(rng, n)
    from s2anet_amd.rotated import nms_rotated_raw
    assert np.array_equal(nms_rotated_raw(cu(d), cu(s), 0.5).cpu().numpy(), oracle.nms_rotated(d, s, 0.5, cull=True))


def test_nms_float64_dispatch_matches_reference_double_arithmetic(rng):
    """float64 boxes take the double instantiation (the reference dispatches on dets' dtype, nms_rotated_cuda.cu:95-100):
    on the fixture built so that 136 keep decisions differ between float32 and float64 arithmetic, the GPU keep lists
    equal the reference's CPU op run on the double tensors; fresh inputs against the (pinned) double oracle with the GPU
    rule; float32 inputs still take the float32 path."""
    import s2anet_amd as S
    g = golden("nms_f64.npz")
    d, s, l, thr = g["dets"], g["scores"], g["labels"], float(g["thr"])
    td, ts, tl = torch.from_numpy(d).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(l).cuda()
    k_ml = S.ml_nms_rotated(td, ts, tl, thr).cpu().numpy()
    from s2anet_amd.rotated import nms_rotated_raw
    k_sc = nms_rotated_raw(td, ts, thr).cpu().numpy()
    # (no pair of the fixture sits EXACTLY on the threshold, so the CPU op's >= and the GPU op's > agree)
    assert np.array_equal(k_ml, g["ml_keep_f64"]) and np.array_equal(k_sc, g["sc_keep_f64"])
    k32 = S.ml_nms_rotated(td.float(), ts.float(), tl.float(), thr).cpu().numpy()
    assert not np.array_equal(k32, g["ml_keep_f64"])                  # float32 inputs: the float32 evaluation, as the reference
    for n, thr2 in ((1, 0.5), (65, 0.3), (3000, 0.5)):
        dd = rand_rboxes(rng, n, span=400).astype(np.float64)
        ss = distinct_scores(rng, n).astype(np.float64)
        ll = rng.integers(0, 6, n).astype(np.float64)
        got = S.ml_nms_rotated(torch.from_numpy(dd).cuda(), torch.from_numpy(ss).cuda(), torch.from_numpy(ll).cuda(), thr2)
        assert np.array_equal(got.cpu().numpy(), oracle.nms_rotated_f64(dd, ss, thr2, labels=ll))
    empty = S.ml_nms_rotated(torch.zeros((0, 5), dtype=torch.float64).cuda(), torch.zeros(0, dtype=torch.float64).cuda(),
                             torch.zeros(0, dtype=torch.float64).cuda(), 0.5)
    assert empty.numel() == 0 and empty.dtype == torch.int64


# ------------------------------------------------------------------ ORN
def test_arf_and_pool(rng):
    import s2anet_amd as S
    g = golden("arf_small.npz")
    for tag in ("s1", "s8"):
        out = S.arf_forward(cu(g[f"w_{tag}"]), cu(g[f"idx_{tag}"]))
        assert np.array_equal(out.cpu().numpy(), g[f"out_{tag}"])
    w = rng.standard_normal((32, 256, 1, 3, 3)).astype(np.float32)
    idx = oracle.arf_indices(1, 8, 3)
    out = S.arf_forward(cu(w), cu(idx))
    assert out.shape == (256, 256, 3, 3)
    assert np.array_equal(out.cpu().numpy(), oracle.arf_forward(w, idx))
    outh = S.arf_forward(cu(w).half(), cu(idx))
    assert torch.equal(outh, out.half())
    # ORConv2d caches the expansion at inference and follows weight updates
    oc = S.ORConv2d(256, 32, kernel_size=3, padding=1, arf_config=(1, 8)).to(dev()).eval()
    with torch.no_grad():
        a = oc.rotate_arf()
        assert oc.rotate_arf() is a
        oc.weight.mul_(2.0)
        assert torch.equal(oc.rotate_arf(), a * 2)
        x = torch.randn(2, 256, 9, 11, device=dev())
        y = oc(x)
        assert y.shape == (2, 256, 9, 11)
    x = rng.standard_normal((2, 64, 7, 9)).astype(np.float32)
    x[0, 3, 2, 2] = np.nan
    ref = oracle.rot_inv_pool(x, 8)
    pool = S.RotationInvariantPooling(64, 8)
    o = pool(cu(x)).cpu().numpy()
    tref = torch.from_numpy(x).view(2, 8, 8, 7, 9).max(2)[0].numpy()
    assert np.array_equal(o, tref, equal_nan=True)
    m = ~np.isnan(tref)
    assert np.array_equal(o[m], ref[m])
    xc = cu(x).contiguous(memory_format=torch.channels_last)
    oc_ = pool(xc)
    assert oc_.is_contiguous(memory_format=torch.channels_last)
    assert np.array_equal(oc_.cpu().numpy(), tref, equal_nan=True)
    assert torch.equal(pool(cu(x).half()).float().nan_to_num(0), torch.from_numpy(x).half().view(2, 8, 8, 7, 9).max(2)[0].float().nan_to_num(0).to(dev()))


# ------------------------------------------------------------------ head glue
def test_glue_kernels():
    from s2anet_amd import _lib
    from s2anet_amd.alignconv import align_offsets
    from s2anet_amd.head import delta2bbox_rotated, fam_refine_anchors
    g = golden("head_glue.npz")
    for key, clip in (("dec_clip_fam", 1e-6), ("dec_clip_odm", 16 / 1000)):
        o = delta2bbox_rotated(cu(g["dec_anchors"]), cu(g["dec_deltas"]), clip).cpu().numpy()
        assert np.allclose(o, g[key], rtol=1e-5, atol=1e-4), key
    off = align_offsets(cu(g["off_anchors"])[None], (12, 20), 8)[0].cpu().numpy()
    assert np.allclose(off, g["off_s8"], rtol=1e-5, atol=1e-4)
    # fam refine == grid anchors + decode(clip 1e-6) of the NCHW prediction map
    rng = np.random.default_rng(5)
    pred = (rng.standard_normal((2, 5, 12, 20)) * 0.3).astype(np.float32)
    ref = np.stack([oracle.delta2bbox_rotated(g["anchors_s8"], pred[b].transpose(1, 2, 0).reshape(-1, 5), 1e-6)
                    for b in range(2)]).reshape(2, 12, 20, 5)
    o = fam_refine_anchors(cu(pred), 8).cpu().numpy()
    assert np.allclose(o, ref, rtol=1e-5, atol=1e-4)
    o2 = fam_refine_anchors(cu(pred).contiguous(memory_format=torch.channels_last), 8).cpu().numpy()
    assert np.array_equal(o, o2)


# ------------------------------------------------------------------ deformable conv / AlignConv
def _dcn_inputs(rng, B, C, H, W, O, sigma=1.5):
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((O, C, 3, 3)) * 0.05).astype(np.float32)
    off = (rng.standard_normal((B, 18, H, W)) * sigma).astype(np.float32)
    off[0, :, 0, 0] = 40.0
    off[0, :, 1, 1] = -0.999
    return x, w, off


def test_dcn_generic_path_golden():
    import s2anet_amd as S
    g = golden("dcn_small.npz")                       # C=16: generic kernel
    conv = S.DeformConv(16, 8, 3, padding=1).to(dev())
    with torch.no_grad():
        conv.weight.copy_(cu(g["weight"]))
        out = conv(cu(g["x"]), cu(g["offset"])).cpu().numpy()
    assert np.allclose(out, g["out_torch"], rtol=1e-4, atol=1e-4)
    # stride / dilation / groups / deformable groups vs the oracle
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 8, 10, 9)).astype(np.float32)
    w = rng.standard_normal((6, 4, 3, 3)).astype(np.float32)
    off = (rng.standard_normal((2, 36, 4, 4)) * 1.2).astype(np.float32)
    ref = oracle.deform_conv_forward(x, off, w, (2, 2), (1, 1), (2, 2), 2, 2)
    out = S.deform_conv(cu(x), cu(off), cu(w), 2, 1, 2, 2, 2).cpu().numpy()
    assert np.allclose(out, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("kernel", ["x3", "mfma32"])
@pytest.mark.parametrize("shape", [(1, 64, 13, 17, 64), (2, 256, 16, 16, 256), (1, 32, 8, 8, 128), (3, 64, 5, 7, 320)])
def test_dcn_mfma_f32_vs_oracle(rng, shape, kernel, monkeypatch):
    """the fused f32 forward against the oracle, both kernels: the default since round 6 (k_dcn_x3: every f32 operand as
    three bf16 planes, six 16-bit products per f32 product) and the f32 matrix instruction's (S2A_DCN_F32=mfma32); the
    second parameter value also holds the two against each other an order of magnitude inside the bound, on operands
    whose magnitudes span many binades"""
    import s2anet_amd as S
    if kernel == "mfma32":
        monkeypatch.setenv("S2A_DCN_F32", "mfma32")
    else:
        monkeypatch.delenv("S2A_DCN_F32", raising=False)
    B, C, H, W, O = shape
    x, w, off = _dcn_inputs(rng, B, C, H, W, O)
    ref = oracle.deform_conv_forward(x, off, w)
    out = S.deform_conv(cu(x), cu(off), cu(w), 1, 1, 1, 1, 1).cpu().numpy()
    err = np.abs(out - ref).max()
    assert err < 1e-4, err                              # north-star tolerance, f32 path
    if kernel == "mfma32":
        xs = x * np.exp2(rng.integers(-12, 12, (1, C, 1, 1))).astype(np.float32)       # channel scales over 24 binades
        ws = w * np.exp2(-rng.integers(-12, 12, (1, C, 1, 1))).astype(np.float32)      # ... undone by the filter
        a = S.deform_conv(cu(xs), cu(off), cu(ws), 1, 1, 1, 1, 1).cpu().numpy()
        monkeypatch.delenv("S2A_DCN_F32", raising=False)
        b = S.deform_conv(cu(xs), cu(off), cu(ws), 1, 1, 1, 1, 1).cpu().numpy()
        assert np.abs(a - b).max() < 1e-5 * max(1.0, np.abs(a).max()), (np.abs(a - b).max(), np.abs(a).max())
        monkeypatch.setenv("S2A_DCN_F32", "mfma32")
    # channels-last storage in and out: same numbers
    xc = cu(x).contiguous(memory_format=torch.channels_last)
    outc = S.deform_conv(xc, cu(off), cu(w), 1, 1, 1, 1, 1)
    assert outc.is_contiguous(memory_format=torch.channels_last)
    assert np.abs(outc.cpu().numpy() - ref).max() < 1e-4
    # zero offsets == plain convolution
    z = S.deform_conv(cu(x), cu(off) * 0, cu(w), 1, 1, 1, 1, 1)
    zr = torch.nn.functional.conv2d(cu(x).double(), cu(w).double(), padding=1).float()
    assert (z - zr).abs().max().item() < 1e-4


def test_dcn_f16_vs_oracle(rng):
    import s2anet_amd as S
    B, C, H, W, O = 2, 128, 12, 10, 64
    x, w, off = _dcn_inputs(rng, B, C, H, W, O)
    xh, wh = cu(x).half(), cu(w).half()
    ref = oracle.deform_conv_forward(xh.float().cpu().numpy(), off, wh.float().cpu().numpy(), f16_cols=True)
    out = S.deform_conv(xh, cu(off), wh, 1, 1, 1, 1, 1)
    assert out.dtype == torch.float16
    err = (out.float().cpu().numpy() - ref)
    assert np.abs(err).max() < 2e-2 and np.abs(err).mean() < 2e-3      # f16 output rounding
    outc = S.deform_conv(xh.contiguous(memory_format=torch.channels_last), cu(off), wh, 1, 1, 1, 1, 1)
    assert np.abs(outc.float().cpu().numpy() - ref).max() < 2e-2


def test_alignconv_fused_vs_oracle(rng):
    """anchors -> offsets (oracle numpy restatement of get_offset) -> deform conv oracle -> ReLU"""
    import s2anet_amd as S
    B, C, H, W, O, stride = 2, 64, 16, 20, 64, 8
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    anchors = np.stack([oracle.grid_anchors(H, W, stride) for _ in range(B)])
    anchors[..., 0:2] += rng.normal(0, 4, anchors[..., 0:2].shape)
    anchors[..., 2:4] = 32 * np.exp(rng.normal(0, 0.5, anchors[..., 2:4].shape))
    anchors[..., 4] = rng.uniform(-np.pi / 4, 3 * np.pi / 4, anchors[..., 4].shape)
    anchors = anchors.astype(np.float32)
    ac = S.AlignConv(C, O, 3).to(dev())
    ac.init_weights()
    w = ac.deform_conv.weight.detach().cpu().numpy() * 5
    with torch.no_grad():
        ac.deform_conv.weight.mul_(5)
        out = ac(cu(x), cu(anchors).view(B, H, W, 5), stride).cpu().numpy()
        off_gpu = torch.stack([ac.get_offset(cu(anchors[b]), (H, W), stride) for b in range(B)])
        unfused = torch.relu(ac.deform_conv(cu(x), off_gpu)).cpu().numpy()
    offs = np.stack([oracle.align_offsets(anchors[b], H, W, stride) for b in range(B)])
    assert np.allclose(off_gpu.cpu().numpy(), offs, rtol=1e-5, atol=1e-4)
    ref = oracle.deform_conv_forward(x, offs, w, relu=True)
    assert np.abs(out - ref).max() < 1e-4
    assert np.abs(unfused - ref).max() < 1e-4
    assert (out >= 0).all() and (out == 0).mean() > 0.2


def test_error_conventions_on_gpu():
    """misuse raises what the reference raises: RuntimeError for shape_check / TORCH_CHECK failures
    (deform_conv_cuda.cpp:62-150, box_iou_rotated.h, nms_rotated_cuda.cu:80-82), ValueError for non-4-D input,
    AssertionError for a batch not divisible by the im2col step (deform_conv.py:26-29,58-63)"""
    import s2anet_amd as S
    conv = S.DeformConv(8, 16, 3, padding=1).to(dev())
    x = torch.zeros(2, 8, 10, 12, device=dev())
    with pytest.raises(RuntimeError):
        conv(x, torch.zeros(2, 16, 10, 12, device=dev()))                 # offset channels != 2 * 9 * groups
    with pytest.raises(RuntimeError):
        conv(x, torch.zeros(2, 18, 9, 12, device=dev()))                  # offset spatial size != output size
    with pytest.raises(RuntimeError):
        conv(torch.zeros(2, 6, 10, 12, device=dev()), torch.zeros(2, 18, 10, 12, device=dev()))   # input planes
    with pytest.raises(RuntimeError):
        conv(x, torch.zeros(1, 18, 10, 12, device=dev()))                 # batch of offset
    with pytest.raises(ValueError):
        S.deform_conv(x[0], torch.zeros(1, 18, 10, 12, device=dev()), conv.weight)
    with pytest.raises(AssertionError):
        S.deform_conv(torch.zeros(3, 8, 10, 12, device=dev()), torch.zeros(3, 18, 10, 12, device=dev()), conv.weight,
                      1, 1, 1, 1, 1, 2)                                  # 3 % im2col_step(2) != 0
    b5 = torch.rand(7, 5, device=dev()) + 1
    with pytest.raises(RuntimeError):
        S.box_iou_rotated(b5[:, :4], b5)
    with pytest.raises(RuntimeError):
        S.ml_nms_rotated(b5, torch.rand(6, device=dev()), torch.zeros(7, device=dev()), 0.5)
    with pytest.raises(RuntimeError):
        S.ml_nms_rotated(b5, torch.rand(7, device=dev()), torch.zeros(5, device=dev()), 0.5)
    assert S.box_iou_rotated(b5[:0], b5).shape == (0, 7)                  # empty inputs are not errors
    assert S.ml_nms_rotated(b5[:0], b5[:0, 0], b5[:0, 0], 0.5).shape == (0,)


def test_fused_conv_epilogue(rng):
    from s2anet_amd.fused import FusedConv2d, bias_act_
    for dt, tol in ((torch.float16, 2e-3), (torch.float32, 1e-6)):
        y = torch.randn(2, 64, 9, 7, device=dev()).to(dt).contiguous(memory_format=torch.channels_last)
        r = torch.randn_like(y)
        b = torch.randn(64, device=dev()).to(dt)
        ref = torch.relu(y.float() + b.float().view(1, -1, 1, 1) + r.float())
        out = bias_act_(y.clone(memory_format=torch.channels_last), b, r, True)
        assert (out.float() - ref).abs().max().item() <= tol * 10
        ref2 = y.float() + b.float().view(1, -1, 1, 1)
        out2 = bias_act_(y.clone(memory_format=torch.channels_last), b, None, False)
        assert (out2.float() - ref2).abs().max().item() <= tol * 10
    conv = torch.nn.Conv2d(16, 32, 3, padding=1).to(dev())
    fc = FusedConv2d.from_conv(conv, relu=True)
    x = torch.randn(2, 16, 8, 8, device=dev()).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert (fc(x) - torch.relu(conv(x))).abs().max().item() < 1e-5


def test_alignconv_f16_large_tile_variants(rng):
    """the 128-position / wave-specialised kernels (picked when >= 512 tiles): f16, both layouts,
    against the f16-column oracle on a random subset of positions (full oracle would take minutes)"""
    import s2anet_amd as S
    from s2anet_amd.alignconv import align_conv_forward
    B, C, H, W, O, stride = 4, 128, 128, 128, 128, 8
    x = (rng.standard_normal((B, C, H, W)) * 0.5).astype(np.float32)
    anchors = np.stack([oracle.grid_anchors(H, W, stride) for _ in range(B)]).reshape(B, H, W, 5).copy()
    anchors[..., 0:2] += rng.normal(0, 4, anchors[..., 0:2].shape)
    anchors[..., 2:4] = 32 * np.exp(rng.normal(0, 0.5, anchors[..., 2:4].shape))
    anchors[..., 4] = rng.uniform(-np.pi / 4, 3 * np.pi / 4, anchors[..., 4].shape)
    anchors = anchors.astype(np.float32)
    w = (rng.standard_normal((O, C, 3, 3)) * 0.05).astype(np.float32)
    xh, wh = cu(x).half(), cu(w).half()
    outs = {}
    for name, xin in (("nchw", xh), ("nhwc", xh.contiguous(memory_format=torch.channels_last))):
        outs[name] = align_conv_forward(xin, cu(anchors), wh, stride, relu=True).float().cpu().numpy()
    assert np.abs(outs["nchw"] - outs["nhwc"]).max() == 0.0       # layout only changes storage
    # oracle on 3 image rows (full rows so offsets index correctly): crop trick does not apply to a
    # deformable op, so evaluate the oracle on the full image 0 but only compare a few rows
    offs = oracle.align_offsets(anchors[0].reshape(-1, 5), H, W, stride)[None]
    ref = oracle.deform_conv_forward(xh[:1].float().cpu().numpy(), offs, wh.float().cpu().numpy(), f16_cols=True, relu=True)
    err = np.abs(outs["nchw"][0] - ref[0])
    assert err.max() < 3e-2 and err.mean() < 2e-3, (err.max(), err.mean())


@pytest.mark.parametrize("shape", [(2, 256, 128, 128, 256), (1, 64, 20, 37, 128), (3, 128, 8, 8, 64), (1, 256, 5, 3, 320)])
def test_own_conv3x3_f16_vs_torch(shape):
    """the patch-staged MFMA 3x3 convolution of the head towers against torch (f32 math on the same
    f16 inputs); bias + ReLU fused; out-channel groupings 4 / 2 / 1 and a ragged last group"""
    from s2anet_amd.fused import conv_f16, conv_pack_weight, FusedConv2d
    B, C, H, W, O = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, C, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(O, C, 3, 3, generator=g) * 0.03).to(dev()).half()
    b = torch.randn(O, generator=g).to(dev()).half()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b.float(), padding=1)
    wp = conv_pack_weight(w)
    for relu in (False, True):
        out = conv_f16(x, wp, b, O, 3, 1, relu)
        r = torch.relu(ref) if relu else ref
        assert out.is_contiguous(memory_format=torch.channels_last) and out.shape == r.shape
        err = (out.float() - r).abs()
        assert err.max().item() < 2e-2 and err.mean().item() < 2e-3, (err.max().item(), err.mean().item())
    out_nb = conv_f16(x, wp, None, O, 3, 1, False)
    assert (out_nb.float() - torch.nn.functional.conv2d(x.float(), w.float(), None, padding=1)).abs().max().item() < 2e-2
    conv = torch.nn.Conv2d(C, O, 3, padding=1).to(dev()).half()
    fc = FusedConv2d.from_conv(conv, relu=True)
    with torch.no_grad():
        assert (fc(x).float() - torch.relu(conv(x)).float()).abs().max().item() < 3e-2


@pytest.mark.parametrize("shape", [(2, 64, 64, 64, 256, 1), (2, 256, 33, 47, 64, 1), (1, 512, 32, 32, 128, 1),
                                   (2, 256, 64, 64, 512, 2), (1, 1024, 17, 9, 2048, 2), (4, 128, 40, 40, 320, 1)])
def test_own_conv1x1_f16_vs_torch(shape):
    """1x1 convolutions (bottleneck / lateral layers) on the same pipeline: stride 1 and 2, residual add"""
    from s2anet_amd.fused import conv_f16, conv_pack_weight
    B, C, H, W, O, st = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, C, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(O, C, 1, 1, generator=g) * 0.05).to(dev()).half()
    b = torch.randn(O, generator=g).to(dev()).half()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b.float(), stride=st)
    res = torch.randn(ref.shape, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
    wp = conv_pack_weight(w)
    out = conv_f16(x, wp, b, O, 1, st, True)
    assert out.shape == ref.shape
    err = (out.float() - torch.relu(ref)).abs()
    assert err.max().item() < 3e-2 and err.mean().item() < 3e-3, (err.max().item(), err.mean().item())
    out2 = conv_f16(x, wp, b, O, 1, st, True, res)
    err2 = (out2.float() - torch.relu(ref + res.float())).abs()
    assert err2.max().item() < 3e-2 and err2.mean().item() < 3e-3
    out3 = conv_f16(x, wp, None, O, 1, st, False, res)
    assert (out3.float() - (torch.nn.functional.conv2d(x.float(), w.float(), None, stride=st) + res.float())).abs().max().item() < 3e-2


def test_own_conv3x3_stride2_vs_torch():
    """the down-sampling 3x3 / stride 2 / pad 1 convolutions of the trunk on the own kernel (4 x 16 output tiles
    from 9 x 33-pixel patches), odd and even sizes, both out-channel groupings"""
    from s2anet_amd.fused import conv_f16, conv_pack_weight
    g = torch.Generator().manual_seed(13)
    for (B, C, H, W, O) in ((2, 128, 64, 96, 128), (1, 256, 37, 51, 256), (3, 64, 10, 18, 384), (2, 512, 16, 16, 512)):
        x = torch.randn(B, C, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(O, C, 3, 3, generator=g) * 0.03).to(dev()).half()
        b = torch.randn(O, generator=g).to(dev()).half()
        ref = torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), b.float(), stride=2, padding=1))
        out = conv_f16(x, conv_pack_weight(w), b, O, 3, 2, True)
        assert out.shape == ref.shape
        err = (out.float() - ref).abs()
        assert err.max().item() < 3e-2 and err.mean().item() < 3e-3, ((B, C, H, W, O), err.max().item())


def test_narrow_conv3x3_filter_through_lds_matches(monkeypatch):
    """64- and 128-map 3x3 layers on 16 x 16 tiles with the filter staged through LDS (S2A_CONV_PH_NARROW=2) are
    bit-identical to the 8 x 16 form, ragged sizes included; same for the fused bottleneck tail"""
    from s2anet_amd.fused import FusedConv2d, bottleneck_tail, conv_f16, conv_pack_weight
    g = torch.Generator().manual_seed(5)
    for (B, C, H, W, O) in ((2, 64, 40, 56, 64), (1, 128, 33, 47, 128), (2, 64, 16, 16, 128), (1, 192, 20, 20, 64)):
        x = torch.randn(B, C, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(O, C, 3, 3, generator=g) * 0.04).to(dev()).half()
        b = torch.randn(O, generator=g).to(dev()).half()
        r = torch.randn(B, O, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        wp = conv_pack_weight(w)
        outs = {}
        for ph in ("1", "2"):
            monkeypatch.setenv("S2A_CONV_PH_NARROW", ph)
            outs[ph] = (conv_f16(x, wp, b, O, 3, 1, True), conv_f16(x, wp, b, O, 3, 1, True, r))
        assert torch.equal(outs["1"][0], outs["2"][0]) and torch.equal(outs["1"][1], outs["2"][1]), (B, C, H, W, O)
        ref = torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), b.float(), padding=1))
        assert (outs["2"][0].float() - ref).abs().max().item() < 3e-2
    c2 = FusedConv2d(64, 64, 3, padding=1, relu=True).to(dev()).half()
    c3 = FusedConv2d(64, 256, 1, relu=True).to(dev()).half()
    x = torch.randn(2, 64, 37, 51, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
    res = torch.randn(2, 256, 37, 51, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        monkeypatch.setenv("S2A_CONV_PH_NARROW", "1")
        a = bottleneck_tail(x, c2, c3, res)
        monkeypatch.setenv("S2A_CONV_PH_NARROW", "2")
        b2 = bottleneck_tail(x, c2, c3, res)
    assert torch.equal(a, b2)


def test_conv1x1_half_tiles_match(monkeypatch):
    """1x1 layers on 64-position tiles (small maps: more workgroups per CU) == the 128-position form, bit for bit"""
    from s2anet_amd.fused import conv_f16, conv_pack_weight
    g = torch.Generator().manual_seed(9)
    for (B, C, H, W, O, st) in ((2, 1024, 24, 40, 256, 1), (1, 256, 33, 47, 1024, 1), (2, 512, 32, 32, 1024, 2), (3, 64, 5, 9, 256, 1)):
        x = torch.randn(B, C, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(O, C, 1, 1, generator=g) * 0.04).to(dev()).half()
        b = torch.randn(O, generator=g).to(dev()).half()
        Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
        r = torch.randn(B, O, Ho, Wo, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        wp = conv_pack_weight(w)
        outs = {}
        for half in ("0", "1"):
            monkeypatch.setenv("S2A_CONV1_HALF", half)
            outs[half] = (conv_f16(x, wp, b, O, 1, st, False), conv_f16(x, wp, b, O, 1, st, True, r))
        assert torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1]), (B, C, H, W, O, st)
        ref = torch.nn.functional.conv2d(x.float(), w.float(), b.float(), stride=st)
        assert (outs["1"][0].float() - ref).abs().max().item() < 6e-2


def test_conv3x3_wide_layer_filter_through_lds_matches(monkeypatch):
    """single-level 256-map 3x3 on 16 x 16 tiles with the filter through LDS (S2A_CONV_PH=2) == 8 x 16 tiles (both on
    16x16x32 MFMAs with the same accumulation order)"""
    from s2anet_amd.fused import conv_f16, conv_pack_weight
    g = torch.Generator().manual_seed(12)
    for (B, C, H, W, O) in ((1, 256, 40, 72, 256), (2, 128, 33, 31, 512)):
        x = torch.randn(B, C, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(O, C, 3, 3, generator=g) * 0.03).to(dev()).half()
        b = torch.randn(O, generator=g).to(dev()).half()
        r = torch.randn(B, O, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        wp = conv_pack_weight(w)
        outs = {}
        for ph in ("1", "2"):
            monkeypatch.setenv("S2A_CONV_PH", ph)
            monkeypatch.setenv("S2A_CONV3_HALF", "0")
            outs[ph] = (conv_f16(x, wp, b, O, 3, 1, True), conv_f16(x, wp, b, O, 3, 1, True, r))
        assert torch.equal(outs["1"][0], outs["2"][0]) and torch.equal(outs["1"][1], outs["2"][1]), (B, C, H, W, O)


def test_conv3x3_half_tiles_match(monkeypatch):
    """3x3 / stride 1 layers on 4 x 16 tiles (small maps) == the 8 x 16 form, bit for bit; with residual; ragged sizes"""
    from s2anet_amd.fused import conv_f16, conv_pack_weight
    g = torch.Generator().manual_seed(10)
    for (B, C, H, W, O) in ((2, 256, 24, 40, 256), (1, 512, 13, 19, 512), (2, 128, 17, 16, 128), (1, 64, 9, 33, 384)):
        x = torch.randn(B, C, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(O, C, 3, 3, generator=g) * 0.03).to(dev()).half()
        b = torch.randn(O, generator=g).to(dev()).half()
        r = torch.randn(B, O, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        wp = conv_pack_weight(w)
        outs = {}
        monkeypatch.setenv("S2A_CONV_PH_NARROW", "1")
        for half in ("0", "1"):
            monkeypatch.setenv("S2A_CONV3_HALF", half)
            outs[half] = (conv_f16(x, wp, b, O, 3, 1, True), conv_f16(x, wp, b, O, 3, 1, False, r))
        assert torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1]), (B, C, H, W, O)
        ref = torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), b.float(), padding=1))
        assert (outs["1"][0].float() - ref).abs().max().item() < 6e-2


def test_bottleneck_tail_fused_matches_two_launches(monkeypatch):
    """conv2 (3x3 64->64) + conv3 (1x1 64->256) + residual + ReLU of a layer-1 bottleneck in one launch
    (s2a_conv3x3_tail1x1_f16): bit-identical to the two stand-alone launches, close to torch fp32; ragged sizes
    (partial tiles), with and without a residual; the detector block takes the fused route"""
    from s2anet_amd.fused import FusedConv2d, bottleneck_tail, bottleneck_tail_ok, conv_f16
    from s2anet_amd.detector import BottleNeck
    g = torch.Generator().manual_seed(21)
    for (B, H, W, with_res) in ((2, 64, 96, True), (1, 37, 51, True), (3, 9, 17, False), (1, 128, 128, True)):
        x = torch.randn(B, 64, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        c2 = FusedConv2d(64, 64, 3, padding=1, relu=True).to(dev()).half()
        c3 = FusedConv2d(64, 256, 1, relu=True).to(dev()).half()
        with torch.no_grad():
            c2.weight.copy_(torch.randn(64, 64, 3, 3, generator=g) * 0.05); c2.bias.copy_(torch.randn(64, generator=g) * 0.1)
            c3.weight.copy_(torch.randn(256, 64, 1, 1, generator=g) * 0.1); c3.bias.copy_(torch.randn(256, generator=g) * 0.1)
            res = (torch.randn(B, 256, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
                   if with_res else None)
            assert bottleneck_tail_ok(x, c2, c3, res) or B * ((H + 7) // 8) * ((W + 15) // 16) < 64
            fused = bottleneck_tail(x, c2, c3, res)
            w2, b2, _ = c2.packed_args(); w3, b3, _ = c3.packed_args()
            two = conv_f16(conv_f16(x, w2, b2, 64, 3, 1, True), w3, b3, 256, 1, 1, True, res)
            assert torch.equal(fused, two), (B, H, W, (fused.float() - two.float()).abs().max().item())
            m = torch.relu(torch.nn.functional.conv2d(x.float(), c2.weight.float(), c2.bias.float(), padding=1)).half().float()
            ref = torch.nn.functional.conv2d(m, c3.weight.float(), c3.bias.float())
            ref = torch.relu(ref + (res.float() if with_res else 0))
            err = (fused.float() - ref).abs()
            assert err.max().item() < 3e-2 and err.mean().item() < 2e-3, (B, H, W, err.max().item())
    # the next block's conv1 chained onto the tail kernel: both outputs bit-identical to the separate launches
    c1n = FusedConv2d(256, 64, 1, relu=True).to(dev()).half()
    for (B, H, W) in ((2, 64, 96), (1, 37, 51), (1, 128, 128)):
        x = torch.randn(B, 64, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        res = torch.randn(B, 256, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            c1n.weight.copy_(torch.randn(256 * 64, generator=g).view(64, 256, 1, 1) * 0.05)
            c1n.bias.copy_(torch.randn(64, generator=g) * 0.1)
            for ph in ("1", "2"):
                monkeypatch.setenv("S2A_CONV_PH_NARROW", ph)
                y, m1 = bottleneck_tail(x, c2, c3, res, c1n)
                y0 = bottleneck_tail(x, c2, c3, res)
                wc, bc, _ = c1n.packed_args()
                m0 = conv_f16(y0, wc, bc, 64, 1, 1, True)
                assert torch.equal(y, y0) and torch.equal(m1, m0), (B, H, W, ph, (m1.float() - m0.float()).abs().max().item())
                # 128 maps: the first block of the next stage
                c1w = FusedConv2d(256, 128, 1, relu=True).to(dev()).half()
                c1w.weight.copy_(torch.randn(128 * 256, generator=g).view(128, 256, 1, 1) * 0.05)
                c1w.bias.copy_(torch.randn(128, generator=g) * 0.1)
                y2, m2 = bottleneck_tail(x, c2, c3, res, c1w)
                ww, bw, _ = c1w.packed_args()
                assert torch.equal(y2, y0) and torch.equal(m2, conv_f16(y0, ww, bw, 128, 1, 1, True)), (B, H, W, ph)
        monkeypatch.delenv("S2A_CONV_PH_NARROW")
    # the detector's block: fused route == separate launches
    blk = BottleNeck(256, 64).to(dev())
    with torch.no_grad():
        for bn in (blk.bn1, blk.bn2, blk.bn3):
            bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5)
    blk.eval().half()
    # fold by hand (fold_batchnorm expects a whole detector)
    def fold(conv, bn):
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        f = torch.nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=True).to(dev()).half()
        f.weight.data = conv.weight * scale.view(-1, 1, 1, 1); f.bias.data = (0 - bn.running_mean) * scale + bn.bias
        return FusedConv2d.from_conv(f, relu=True)
    blk.conv1, blk.bn1 = fold(blk.conv1, blk.bn1), torch.nn.Identity()
    blk.conv2, blk.bn2 = fold(blk.conv2, blk.bn2), torch.nn.Identity()
    blk.conv3, blk.bn3 = fold(blk.conv3, blk.bn3), torch.nn.Identity()
    x = torch.randn(2, 256, 64, 64, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        a = blk(x)
        monkeypatch.setenv("S2A_NO_FUSED_TAIL", "1")
        b = blk(x)
    assert torch.equal(a, b)
    monkeypatch.delenv("S2A_NO_FUSED_TAIL")
    # a stage of two such blocks: the first block's tail hands the second its conv1 output
    from s2anet_amd.detector import DetectorBackbone
    import copy
    blk2 = copy.deepcopy(blk)
    with torch.no_grad():
        blk2.conv1.weight.mul_(0.7); blk2.conv3.bias.add_(0.05)
        seq = torch.nn.Sequential(blk, blk2)
        a = DetectorBackbone.run_blocks(seq, x)
        monkeypatch.setenv("S2A_NO_TAIL_CHAIN", "1")
        b = DetectorBackbone.run_blocks(seq, x)
        c = seq(x)
    assert torch.equal(a, b) and torch.equal(a, c)
    monkeypatch.delenv("S2A_NO_TAIL_CHAIN")
    # the whole trunk: chains inside layer1 and across the layer1 -> layer2 boundary == no chaining
    from s2anet_amd.detector import build_synthetic_detector
    model = build_synthetic_detector(num_classes=15, seed=3, dtype=torch.float16, device=dev())
    imgs = torch.randint(0, 256, (2, 3, 512, 512), dtype=torch.uint8, generator=g).to(dev()).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert model.backbone.stem_fusable(imgs)
        f1 = model.backbone.forward_u8(imgs)
        monkeypatch.setenv("S2A_NO_TAIL_CHAIN", "1")
        f0 = model.backbone.forward_u8(imgs)
    # C3 = layer2's output: every layer up to there runs on the own (deterministic) kernels at this size; the deeper
    # stages fall to library kernels on such small maps, which are not run-to-run bit-stable
    assert len(f1) == 3 and torch.equal(f1[0], f0[0])
    assert all((p.float() - q.float()).abs().max().item() < 2e-2 for p, q in zip(f1[1:], f0[1:]))


@pytest.mark.parametrize("shape", [(2, 256, 16, 16, 15, 3), (8, 256, 8, 8, 5, 3), (2, 256, 40, 24, 5, 1), (1, 256, 128, 128, 15, 1)])
def test_narrow_prediction_heads(shape):
    """5 / 15-map prediction heads (head.py:205-222) on the own conv kernel with zero-padded filters;
    the result is the [:, :O] view of a 64-channel buffer"""
    from s2anet_amd.fused import FusedConv2d
    B, C, H, W, O, k = shape
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(C, O, k, padding=k // 2).to(dev()).half()
    conv.weight.data.normal_(0, 0.02)
    conv.bias.data.normal_(0, 1.0)
    x = torch.randn(B, C, H, W, device=dev()).half().contiguous(memory_format=torch.channels_last)
    fc = FusedConv2d.from_conv(conv)
    with torch.no_grad():
        y = fc(x)
        ref = torch.nn.functional.conv2d(x.float(), conv.weight.float(), conv.bias.float(), padding=k // 2)
    assert y.shape == ref.shape and hasattr(fc, "_packed")           # own kernel ran (no library fallback)
    assert (y.float() - ref).abs().max().item() < 2e-2


def test_arf_backward_and_autograd(rng):
    import s2anet_amd as S
    g = golden("arf_backward_small.npz")
    for tag in ("s1", "s8"):
        out = S.arf_backward(cu(g[f"idx_{tag}"]), cu(g[f"gout_{tag}"]))
        assert np.array_equal(out.cpu().numpy(), g[f"gin_{tag}"])          # bit-exact vs the reference CPU op
    idx = oracle.arf_indices(1, 8, 3)
    gy = rng.standard_normal((256, 256, 3, 3)).astype(np.float32)
    assert np.array_equal(S.arf_backward(cu(idx), cu(gy)).cpu().numpy(), oracle.arf_backward(idx, gy))
    # autograd through ORConv2d (training-mode filter expansion): gradient reaches the ARF bank
    oc = S.ORConv2d(16, 2, kernel_size=3, padding=1, arf_config=(1, 8)).to(dev()).train()
    x = torch.randn(2, 16, 6, 6, device=dev())
    y = oc(x)
    y.square().sum().backward()
    assert oc.weight.grad is not None and oc.weight.grad.shape == oc.weight.shape
    w_exp = S.arf_forward(oc.weight.detach(), oc.indices).requires_grad_(True)
    y2 = torch.nn.functional.conv2d(x, w_exp, oc.bias, padding=1)
    y2.square().sum().backward()
    assert torch.allclose(oc.weight.grad, S.arf_backward(oc.indices, w_exp.grad), rtol=1e-5, atol=1e-5)


def test_arf_float64_dispatch(rng):
    """arf_forward / arf_backward on float64 tensors (AT_DISPATCH_FLOATING_TYPES, ActiveRotatingFilter_cuda.cu:104,149):
    bit-exact against the reference's CPU op run on the same double tensors (small shapes: its uint16 index wraps beyond
    65 535 elements) and against the definition at the head's shape"""
    import s2anet_amd as S
    from oracle import ref
    orn = ref.orn()
    for shp in ((4, 2, 1, 3, 3), (4, 2, 8, 3, 3)):
        O, I, nOri, kH, kW = shp
        idx = oracle.arf_indices(nOri, 8, 3)
        w = rng.standard_normal(shp)
        g = rng.standard_normal((O * 8, I * nOri, kH, kW))
        out = S.arf_forward(torch.from_numpy(w).cuda(), cu(idx))
        gin = S.arf_backward(cu(idx), torch.from_numpy(g).cuda())
        assert out.dtype == torch.float64 and gin.dtype == torch.float64
        if orn is not None:
            assert np.array_equal(out.cpu().numpy(), orn.arf_forward(torch.from_numpy(w), torch.from_numpy(idx)).numpy())
            assert np.array_equal(gin.cpu().numpy(), orn.arf_backward(torch.from_numpy(idx), torch.from_numpy(g)).numpy())
        # the same values as the float32 path where float32 is exact: forward is a pure copy
        assert np.array_equal(out.cpu().numpy().astype(np.float32), S.arf_forward(cu(w.astype(np.float32)), cu(idx)).cpu().numpy())
    idx = oracle.arf_indices(1, 8, 3)
    w = rng.standard_normal((32, 256, 1, 3, 3))
    got = S.arf_forward(torch.from_numpy(w).cuda(), cu(idx)).cpu().numpy()
    assert np.array_equal(got.astype(np.float32), oracle.arf_forward(w.astype(np.float32), idx)) and got.dtype == np.float64
    assert np.array_equal(got, np.ascontiguousarray(got))          # (values are copies of w: exact in float64 as well)


def test_modulated_deform_conv_forward_vs_torch_formulation(rng):
    """deform_conv_cuda.modulated_deform_conv_cuda_forward (DCNv2): no runnable reference (CUDA only) -> checked against
    an independent torch formulation (explicit 4-corner gather x mask, einsum with the grouped filter, + bias) in
    float64; groups, deformable groups, stride 2 / dilation 2 / 5x3 kernels; mask == 1 and bias == 0 reproduce the
    plain deformable convolution"""
    from s2anet_amd.dcn import modulated_deform_conv_cuda_forward, deform_conv_forward_cuda

    def ref(x, w, b, off, msk, stride, pad, dil, group, dg):
        B, C, H, W = x.shape
        O, Cg, kh, kw = w.shape
        Ho = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
        Wo = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
        xd, cols = x.double(), []
        ys = torch.arange(Ho, dtype=torch.float64).view(1, Ho, 1) * stride[0] - pad[0]
        xs = torch.arange(Wo, dtype=torch.float64).view(1, 1, Wo) * stride[1] - pad[1]
        cpg = C // dg
        for c in range(C):
            g_ = c // cpg
            taps = []
            for i in range(kh):
                for j in range(kw):
                    t = i * kw + j
                    hy = ys + i * dil[0] + off[:, g_ * 2 * kh * kw + 2 * t].double()
                    wx = xs + j * dil[1] + off[:, g_ * 2 * kh * kw + 2 * t + 1].double()
                    ok = (hy > -1) & (wx > -1) & (hy < H) & (wx < W)
                    h0, w0 = torch.floor(hy), torch.floor(wx)
                    lh, lw = hy - h0, wx - w0
                    val = torch.zeros_like(hy)
                    for (dy, dx, wt) in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
                        yy, xx = (h0 + dy).long(), (w0 + dx).long()
                        inside = (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
                        v = xd[:, c][torch.arange(B).view(B, 1, 1), yy.clamp(0, H - 1), xx.clamp(0, W - 1)]
                        val = val + torch.where(inside, wt * v, torch.zeros_like(v))
                    taps.append(torch.where(ok, val, torch.zeros_like(val)) * msk[:, g_ * kh * kw + t].double())
            cols.append(torch.stack(taps, 1))                      # [B, kh*kw, Ho, Wo]
        col = torch.stack(cols, 1)                                 # [B, C, kh*kw, Ho, Wo]
        Og = O // group
        out = torch.zeros(B, O, Ho, Wo, dtype=torch.float64)
        for g_ in range(group):
            out[:, g_ * Og:(g_ + 1) * Og] = torch.einsum("ock,bckhw->bohw", w[g_ * Og:(g_ + 1) * Og].double().reshape(Og, Cg, kh * kw),
                                                         col[:, g_ * Cg:(g_ + 1) * Cg])
        return out + (b.double().view(1, -1, 1, 1) if b is not None else 0)

    g = torch.Generator().manual_seed(4)
    cases = [(2, 8, 9, 11, 6, (3, 3), (1, 1), (1, 1), (1, 1), 1, 1, True),
             (1, 8, 12, 10, 8, (3, 3), (2, 2), (1, 1), (1, 1), 2, 2, True),
             (2, 4, 10, 13, 4, (5, 3), (1, 1), (2, 1), (2, 2), 1, 4, False)]
    for (B, C, H, W, O, k, st, pd, dl, group, dg, with_bias) in cases:
        x = torch.randn(B, C, H, W, generator=g)
        w = torch.randn(O, C // group, *k, generator=g) * 0.2
        b = torch.randn(O, generator=g) if with_bias else None
        Ho = (H + 2 * pd[0] - (dl[0] * (k[0] - 1) + 1)) // st[0] + 1
        Wo = (W + 2 * pd[1] - (dl[1] * (k[1] - 1) + 1)) // st[1] + 1
        off = torch.randn(B, dg * 2 * k[0] * k[1], Ho, Wo, generator=g) * 1.5
        msk = torch.rand(B, dg * k[0] * k[1], Ho, Wo, generator=g)
        out = torch.empty(B, O, Ho, Wo, device=dev())
        r = modulated_deform_conv_cuda_forward(cu(x), cu(w), cu(b) if with_bias else torch.empty(0, device=dev()), None,
                                               cu(off), cu(msk), out, None, k[0], k[1], st[0], st[1], pd[0], pd[1],
                                               dl[0], dl[1], group, dg, with_bias)
        assert r is None
        want = ref(x, w, b, off, msk, st, pd, dl, group, dg)
        assert (out.cpu().double() - want).abs().max().item() < 2e-4, (B, C, H, W, O, k)
    # mask of ones, no bias == the plain deformable convolution of the same library
    B, C, H, W, O = 2, 8, 9, 11, 6
    x, w = torch.randn(B, C, H, W, generator=g), torch.randn(O, C, 3, 3, generator=g) * 0.2
    off = torch.randn(B, 18, H, W, generator=g)
    o1, o2 = torch.empty(B, O, H, W, device=dev()), torch.empty(B, O, H, W, device=dev())
    modulated_deform_conv_cuda_forward(cu(x), cu(w), torch.empty(0, device=dev()), None, cu(off), torch.ones(B, 9, H, W, device=dev()),
                                       o1, None, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, False)
    deform_conv_forward_cuda(cu(x), cu(w), cu(off), o2, None, None, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64)
    assert torch.equal(o1, o2)
    with pytest.raises(RuntimeError):
        modulated_deform_conv_cuda_forward(cu(x), cu(w), torch.empty(0, device=dev()), None, cu(off), torch.ones(B, 8, H, W, device=dev()),
                                           o1, None, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, False)


def test_rie_forward_backward(rng):
    """orn_cuda.rie_forward / rie_backward on the GPU: golden (reference CPU op) + oracle on a larger case + autograd"""
    from s2anet_amd.orn import rie_forward, rie_backward, RotationInvariantEncoding
    g = golden("rie_small.npz")
    for tag in ("a", "b"):
        n = int(g[f"n_{tag}"])
        d, al = rie_forward(cu(g[f"f_{tag}"]), n)
        assert d.dtype == torch.uint8 and np.array_equal(d.cpu().numpy(), g[f"dir_{tag}"])
        assert np.array_equal(al.cpu().numpy(), g[f"aligned_{tag}"])
        assert np.array_equal(rie_backward(cu(g[f"dir_{tag}"]), cu(g[f"gout_{tag}"]), n).cpu().numpy(), g[f"gin_{tag}"])
    f = rng.standard_normal((700, 1024 * 8, 1, 1)).astype(np.float32)      # more groups than one grid pass covers
    d, al = rie_forward(cu(f), 8)
    rd, ral = oracle.rie_forward(f, 8)
    assert np.array_equal(d.cpu().numpy(), rd) and np.array_equal(al.cpu().numpy(), ral)
    x = cu(f[:4, :64]).requires_grad_(True)
    y, direction = RotationInvariantEncoding(8, return_direction=True)(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    assert np.array_equal(x.grad.cpu().numpy(), oracle.rie_backward(direction.cpu().numpy(), w.cpu().numpy(), 8))
    with pytest.raises(RuntimeError):
        rie_forward(cu(f[:2, :16, :, 0]), 8)


def test_polyiou_pairs_bitexact(rng):
    from s2anet_amd.rotated import polyiou_pairs
    g = golden("iou_256.npz")
    P1, P2 = oracle.rboxes_to_polys(g["boxes1"]), oracle.rboxes_to_polys(g["boxes2"])
    out = polyiou_pairs(cu(P1[g["poly_i"]]), cu(P2[g["poly_j"]])).cpu().numpy()
    assert np.array_equal(out, g["poly_iou"])                   # f64, bit for bit vs the reference SWIG module
    a, b = oracle.rboxes_to_polys(rand_rboxes(rng, 20000, span=300)), oracle.rboxes_to_polys(rand_rboxes(rng, 20000, span=300))
    b[:10] = a[:10]                                             # identical polygons
    a[10:20, :] = a[10:20, [6, 7, 4, 5, 2, 3, 0, 1]]            # clockwise input -> reversed internally
    out = polyiou_pairs(cu(a), cu(b)).cpu().numpy()
    ref = oracle.polyiou(a, b)
    assert np.array_equal(out, ref, equal_nan=True)
    assert abs(polyiou_pairs(cu(np.array([[0, 0, 1, 0, 1, 1, 0, 1.0]])), cu(np.array([[.5, .5, 1.5, .5, 1.5, 1.5, .5, 1.5]]))).item() - 1 / 7) < 1e-12


@pytest.mark.parametrize("form", ["list", "mask"])
def test_merge_nms_poly(rng, monkeypatch, form):
    """both forms of s2a_nms_poly: the list form (HBB pair list -> polyiou -> suppression edges -> rounds; round 6, the default
    of a synchronous call) and the N x N / 64 mask + scan form (S2A_POLY_NMS_LIST=0; also the overflow fallback of the list)"""
    from s2anet_amd.rotated import nms_poly
    monkeypatch.setenv("S2A_POLY_NMS_LIST", "1" if form == "list" else "0")
    g = golden("merge_nms_poly.npz")
    for thr in (0.1, 0.5):
        k = nms_poly(cu(g["dets"]), thr).cpu().numpy()
        assert np.array_equal(k, g[f"keep_{thr}"])                      # the reference script's own keep list
    n = 5000
    polys = oracle.rboxes_to_polys(rand_rboxes(rng, n, span=700))
    dets = np.concatenate([polys, ((rng.permutation(n) + 1.0) / (n + 1.0))[:, None]], 1)
    assert np.array_equal(nms_poly(cu(dets), 0.3).cpu().numpy(), oracle.nms_poly(dets, 0.3))
    assert nms_poly(torch.zeros((0, 9), device=dev()), 0.5).shape == (0,)
    # dense scene: more HBB-overlapping pairs than the pair list holds -> the direct kernel takes over
    n = 3000
    polys = oracle.rboxes_to_polys(rand_rboxes(rng, n, span=60))
    dets = np.concatenate([polys, ((rng.permutation(n) + 1.0) / (n + 1.0))[:, None]], 1)
    assert np.array_equal(nms_poly(cu(dets), 0.7).cpu().numpy(), oracle.nms_poly(dets, 0.7))
    # negative threshold: every lower-scored box is dropped (hbb_ovr = 0 is not <= thresh), :115
    assert np.array_equal(nms_poly(cu(dets[:500]), -0.1).cpu().numpy(), oracle.nms_poly(dets[:500], -0.1))


def _pyr_setup(B=2, sizes=((40, 56), (20, 28), (10, 14), (5, 7), (3, 4)), C=256):
    from s2anet_amd.pyramid import PyramidLayout
    layout = PyramidLayout(B, sizes, (8, 16, 32, 64, 128))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(layout.pixels, C, generator=g).to(dev()).half()
    return layout, x, g


def test_pyramid_conv3x3_matches_per_level():
    """one pyramid-packed launch == the per-level launches of the same kernel (bit-identical), and == torch"""
    from s2anet_amd import pyramid as P
    from s2anet_amd.fused import conv_f16, conv_pack_weight
    layout, x, g = _pyr_setup()
    for (C, O) in ((256, 256), (256, 64), (32, 256)):
        xx = x[:, :C].contiguous()
        w = (torch.randn(O, C, 3, 3, generator=g) * 0.03).to(dev()).half()
        b = torch.randn(O, generator=g).to(dev()).half()
        wp = conv_pack_weight(w)
        out = P.conv3x3(layout, xx, wp, b, O, relu=True)
        for l in range(len(layout.sizes)):
            xl = layout.level(xx, l)
            ref = torch.relu(torch.nn.functional.conv2d(xl.float(), w.float(), b.float(), padding=1))
            got = layout.level(out, l)
            assert (got.float() - ref).abs().max().item() < 3e-2
            if layout.sizes[l][0] >= 8 and C % 64 == 0:
                assert torch.equal(got, conv_f16(xl.contiguous(memory_format=torch.channels_last), wp, b, O, 3, 1, True))


def test_pyramid_orconv_with_fused_pooling():
    """ORConv + orientation max-pool in one launch == conv launch followed by the pooling kernel, bit for bit"""
    from s2anet_amd import pyramid as P
    from s2anet_amd.fused import conv_pack_weight
    layout, x, g = _pyr_setup()
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev()).half()
    b = torch.randn(256, generator=g).to(dev()).half()
    wp = conv_pack_weight(w)
    ref = P.conv3x3(layout, x, wp, b, 256, relu=False)
    out, pooled = P.orconv_pool(layout, x, wp, b, 256)
    assert torch.equal(out, ref) and torch.equal(pooled, P.rot_inv_pool(ref, 8))
    assert torch.equal(pooled, ref.view(-1, 32, 8).max(dim=2)[0])


def test_pyramid_tower_with_fused_head():
    """last tower conv + 1x1 prediction head in one launch == the two launches (same f16 staging), both head widths"""
    from s2anet_amd import pyramid as P
    from s2anet_amd.fused import conv_pack_weight
    layout, x, g = _pyr_setup()
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev()).half()
    b = torch.randn(256, generator=g).to(dev()).half()
    wp = conv_pack_weight(w)
    tower = P.conv3x3(layout, x, wp, b, 256, relu=True)
    for nh in (5, 15):
        hw = (torch.randn(nh, 256, 1, 1, generator=g) * 0.05).to(dev()).half()
        hb = torch.cat([torch.randn(nh, generator=g), torch.zeros(64 - nh)]).to(dev()).half()
        hwp = conv_pack_weight(hw)
        ref = P.conv1x1(tower, hwp, hb, 64, relu=False)
        got = P.conv3x3_head(layout, x, wp, b, 256, hwp, hb, relu=True)
        assert (got[:, :nh].float() - ref[:, :nh].float()).abs().max().item() < 2e-3      # same operands, f32 sums in another order
        got2, tw = P.conv3x3_head(layout, x, wp, b, 256, hwp, hb, relu=True, keep_tower=True)
        assert torch.equal(tw, tower) and torch.equal(got2[:, :nh], got[:, :nh])


def test_pyramid_alignconv_and_refine(rng):
    """pyramid-packed fam_refine + AlignConv (k_dcn_patch) against the per-level entry points and the oracle"""
    from s2anet_amd import pyramid as P
    from s2anet_amd.alignconv import align_conv_forward, pack_weight
    from s2anet_amd.head import fam_refine_anchors
    layout, x, g = _pyr_setup()
    pred = (torch.randn(layout.pixels, 64, generator=g) * 0.3).to(dev()).half()
    anchors = P.fam_refine_anchors(layout, pred, 4.0)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev()).half()
    out = P.align_conv(layout, x, anchors, pack_weight(w, torch.float16), 256)
    for l, st in enumerate(layout.strides):
        a_ref = fam_refine_anchors(layout.level(pred, l, 5), st, 4.0)
        a_got = layout.rows(anchors, l).view(a_ref.shape)
        assert torch.equal(a_got, a_ref)
        xl = layout.level(x, l).contiguous(memory_format=torch.channels_last)
        ref = align_conv_forward(xl, a_ref, w, st, relu=True)
        got = layout.level(out, l)
        assert (got.float() - ref.float()).abs().max().item() < 3e-2
    # the packed launch against the CPU oracle (f16-rounded columns, f32 sums, on the same f16 inputs): the two LARGE
    # levels (the ones the benchmark's time goes to: many tiles per image, tiles that straddle the image border) and
    # the smallest one (a single partial tile)
    wf = w.float().cpu().numpy()
    for l in (0, 1, 4):
        H, W = layout.sizes[l]
        a = layout.rows(anchors, l).view(layout.batch, H * W, 5).cpu().numpy()
        xl = layout.level(x, l).float().cpu().numpy()
        for bi in range(layout.batch):
            off = oracle.align_offsets(a[bi], H, W, layout.strides[l])
            ref = oracle.deform_conv_forward(np.ascontiguousarray(xl[bi:bi + 1]), off[None], wf, f16_cols=True, relu=True)
            got = layout.level(out, l)[bi:bi + 1].float().cpu().numpy()
            err = np.abs(got - ref)
            assert err.max() < 2e-2 and err.mean() < 2e-3, (l, bi, err.max(), err.mean())


def test_pyramid_alignconv_wild_anchors_vs_oracle(monkeypatch):
    """the pyramid-packed AlignConv launch with more tiles than CUs, ragged level sizes and WILD anchors (bilinear corners
    that leave the 16 x 24 LDS patch take the global-gather path; tame anchors stay inside it): every level of every
    image against the f16-column oracle, and two launches of the same inputs bit-identical (round 2 also held four
    alternative forms of this launch to that -- persistent, three-slot ring, two workgroups per CU, half-tile tail --
    all measured slower and removed, as was round 4's symmetric 16 x 16-tile form k_dcn_sym)"""
    from s2anet_amd import pyramid as P
    from s2anet_amd.alignconv import pack_weight
    B, C = 2, 256
    sizes = [(96, 136), (48, 68), (24, 34), (12, 17), (6, 9)]
    strides = (8, 16, 32, 64, 128)
    lay = P.PyramidLayout(B, sizes, strides)
    g = torch.Generator().manual_seed(79)
    x = torch.relu(torch.randn(lay.pixels, C, generator=g)).to(dev()).half()
    w = (torch.randn(256, C, 3, 3, generator=g) * 0.02).to(dev()).half()
    wp = pack_weight(w, torch.float16)
    wf = w.float().cpu().numpy()
    for jitter, spread in ((0.05, 0.1), (0.9, 0.7)):
        anchors = []
        for (h, ww), st in zip(sizes, strides):
            ys, xs = torch.meshgrid(torch.arange(h), torch.arange(ww), indexing="ij")
            a = torch.stack([xs * st + 0.5 * (st - 1) + torch.randn(h, ww, generator=g) * st * jitter,
                             ys * st + 0.5 * (st - 1) + torch.randn(h, ww, generator=g) * st * jitter,
                             4 * st * torch.exp(torch.randn(h, ww, generator=g) * spread),
                             4 * st * torch.exp(torch.randn(h, ww, generator=g) * spread),
                             torch.rand(h, ww, generator=g) * 3.14159 - 0.785], -1).float()
            anchors.append(a.unsqueeze(0).expand(B, -1, -1, -1).reshape(-1, 5))
        anchors = torch.cat(anchors).to(dev()).contiguous()
        out = P.align_conv(lay, x, anchors, wp, 256).clone()
        assert torch.equal(out, P.align_conv(lay, x, anchors, wp, 256))
        # 304 tiles on 256 CUs: the 48 tiles behind the full round run as 96 half tiles inside the same launch
        # (PatchArgs::n_full); S2A_DCN_HALF_TAIL=0 launches 304 full tiles instead -- the same bits either way
        monkeypatch.setenv("S2A_DCN_HALF_TAIL", "0")
        assert torch.equal(out, P.align_conv(lay, x, anchors, wp, 256))
        monkeypatch.setenv("S2A_DCN_HALF_TAIL", "1")
        assert torch.equal(out, P.align_conv(lay, x, anchors, wp, 256))
        monkeypatch.delenv("S2A_DCN_HALF_TAIL")
        for l in (1, 2, 4):                                   # (level 0 at this size: 20 s of oracle per image)
            H, W = sizes[l]
            a = lay.rows(anchors, l).view(B, H * W, 5).cpu().numpy()
            xl = lay.level(x, l).float().cpu().numpy()
            for bi in range(B):
                off = oracle.align_offsets(a[bi], H, W, strides[l])
                ref = oracle.deform_conv_forward(np.ascontiguousarray(xl[bi:bi + 1]), off[None], wf, f16_cols=True, relu=True)
                err = np.abs(lay.level(out, l)[bi:bi + 1].float().cpu().numpy() - ref)
                assert err.max() < 2e-2 and err.mean() < 2e-3, (jitter, l, bi, err.max(), err.mean())


def test_alignconv_small_grid_half_tiles_match(monkeypatch):
    """a launch with fewer 8 x 16 tiles than 1.5 x CUs (BASELINE configs[1]: one P3 level of one chip) runs as 4 x 16 half
    tiles: bit-identical to the full-tile launch (S2A_DCN_NO_HALF=1), both layouts of the output"""
    from s2anet_amd.alignconv import align_conv_forward
    g = torch.Generator().manual_seed(80)
    B, C, H, W, O, st = 1, 256, 128, 120, 256, 8
    x = torch.randn(B, C, H, W, generator=g).to(dev()).half()
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    anc = torch.stack([xs * st + 3.5 + torch.randn(H, W, generator=g) * 4, ys * st + 3.5 + torch.randn(H, W, generator=g) * 4,
                       32 * torch.exp(torch.randn(H, W, generator=g) * 0.5), 32 * torch.exp(torch.randn(H, W, generator=g) * 0.5),
                       torch.rand(H, W, generator=g) * 3.14159 - 0.785], -1)[None].to(dev())
    w = (torch.randn(O, C, 3, 3, generator=g) * 0.02).to(dev()).half()
    for xin in (x, x.contiguous(memory_format=torch.channels_last)):
        outs = {}
        for mode in ("", "1"):
            if mode:
                monkeypatch.setenv("S2A_DCN_NO_HALF", mode)
            else:
                monkeypatch.delenv("S2A_DCN_NO_HALF", raising=False)
            outs[mode] = align_conv_forward(xin, anc, w, st, relu=True).clone()
        assert torch.equal(outs[""], outs["1"])
        assert outs[""].float().abs().sum().item() > 0


def test_detector_pyramid_path_matches_per_level(monkeypatch):
    """the whole head on the pyramid-packed path == the per-level path (library kernels on the small levels)"""
    from s2anet_amd.detector import build_synthetic_detector
    m = build_synthetic_detector(device=dev())
    g = torch.Generator().manual_seed(1)
    imgs = (torch.rand(2, 3, 384, 512, generator=g)).to(dev()).half().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        got = m.features_to_pred(imgs)
        monkeypatch.setenv("S2A_NO_PYRAMID", "1")
        ref = m.features_to_pred(imgs)
    assert hasattr(m, "_layout")                                   # the packed path ran
    for gl, rl in zip(got, ref):
        for a, b in zip(gl, rl):
            if a is None:
                assert b is None
                continue
            assert a.shape == b.shape
            d = (a.float() - b.float()).abs().max().item()
            assert d < 5e-2 * max(1.0, b.float().abs().max().item()), d


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 200, 132), (3, 39, 44), (1, 256, 256)])
def test_fused_stem_vs_torch(shape):
    """uint8 / 255 -> conv7x7/2 + bias -> ReLU -> maxpool3x3/2 in one kernel against the stock ops"""
    from s2anet_amd.fused import stem_pack_weight, stem_u8
    B, H, W = shape
    g = torch.Generator().manual_seed(2)
    img = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g).to(dev()).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(dev()).half()
    b = torch.randn(64, generator=g).to(dev()).half()
    x = img.half().div_(255.0)                                      # the stock normalisation (f16 division)
    ref = torch.nn.functional.max_pool2d(torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), b.float(), stride=2, padding=3)),
                                         3, 2, 1)
    out = stem_u8(img, stem_pack_weight(w), b)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    err = (out.float() - ref).abs()
    assert err.max().item() < 2e-2 and err.mean().item() < 2e-3, (err.max().item(), err.mean().item())
    out0 = stem_u8(img, stem_pack_weight(w), None)
    ref0 = torch.nn.functional.max_pool2d(torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), None, stride=2, padding=3)), 3, 2, 1)
    assert (out0.float() - ref0).abs().max().item() < 2e-2


def test_detect_fused_stem_matches_stock(monkeypatch):
    from s2anet_amd.detector import build_synthetic_detector
    m = build_synthetic_detector(device=dev())
    g = torch.Generator().manual_seed(4)
    img = torch.randint(0, 256, (2, 3, 384, 384), dtype=torch.uint8, generator=g).to(dev()).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert m.backbone.stem_fusable(img)
        a = m.backbone.forward_u8(img)
        b = m.backbone(img.half().div_(255.0).contiguous(memory_format=torch.channels_last))
    for u, v in zip(a, b):
        assert u.shape == v.shape
        assert (u.float() - v.float()).abs().max().item() < 3e-2 * max(1.0, v.float().abs().max().item())


def test_fpn_topdown_step_fused():
    """conv1x1 + bias + nearest-2x-upsample(coarse) in one launch == the separate ops (same f16 roundings)"""
    from s2anet_amd.fused import conv_f16, conv_pack_weight, conv1x1_add_up2
    g = torch.Generator().manual_seed(9)
    for (B, C, H, W, O) in ((2, 512, 32, 48, 256), (1, 1024, 64, 64, 256), (3, 128, 8, 20, 64)):
        x = torch.randn(B, C, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(O, C, 1, 1, generator=g) * 0.04).to(dev()).half()
        b = torch.randn(O, generator=g).to(dev()).half()
        coarse = torch.randn(B, O, H // 2, W // 2, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
        wp = conv_pack_weight(w)
        got = conv1x1_add_up2(x, wp, b, coarse, O)
        ref = conv_f16(x, wp, b, O, 1, 1, False) + torch.nn.functional.interpolate(coarse, scale_factor=2, mode="nearest")
        assert torch.equal(got, ref)


def test_dcn_backward_vs_oracle_and_golden(rng):
    """deform_conv backward (three HIP kernels + library GEMMs, autograd wiring) against the golden autograd
    values, the oracle on a second shape (stride/dilation/deformable groups), and the pybind-shaped entry points"""
    import s2anet_amd as S
    from s2anet_amd.dcn import deform_conv, deform_conv_backward_input_cuda, deform_conv_backward_parameters_cuda
    g = golden("dcn_backward_small.npz")
    x, off, w = (cu(g[k]).requires_grad_(True) for k in ("x", "offset", "weight"))
    out = deform_conv(x, off, w, 1, 1, 1, 1, 1)
    out.backward(cu(g["grad_out"]))
    for got, ref in ((x.grad, g["grad_input"]), (off.grad, g["grad_offset"]), (w.grad, g["grad_weight"])):
        assert np.abs(got.cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    # general geometry: stride 2, dilation 2, two deformable groups, im2col_step < batch
    B, C, H, W, O, dg = 4, 8, 11, 13, 6, 2
    st, pd, dl = (2, 1), (2, 1), (2, 1)
    Ho = (H + 2 * pd[0] - (dl[0] * 2 + 1)) // st[0] + 1
    Wo = (W + 2 * pd[1] - (dl[1] * 2 + 1)) // st[1] + 1
    xn = rng.standard_normal((B, C, H, W)).astype(np.float32)
    wn = (rng.standard_normal((O, C, 3, 3)) * 0.2).astype(np.float32)
    on = (rng.standard_normal((B, dg * 18, Ho, Wo)) * 1.5).astype(np.float32)
    gn = rng.standard_normal((B, O, Ho, Wo)).astype(np.float32)
    gx, goff, gw = oracle.deform_conv_backward(xn, on, wn, gn, st, pd, dl, dg)
    gi, go_ = torch.zeros(B, C, H, W, device=dev()), torch.zeros(B, dg * 18, Ho, Wo, device=dev())
    args = (3, 3, st[1], st[0], pd[1], pd[0], dl[1], dl[0], 1, dg)
    assert deform_conv_backward_input_cuda(cu(xn), cu(on), cu(gn), gi, go_, cu(wn), None, *args, 2) == 1
    gwt = torch.zeros(O, C, 3, 3, device=dev())
    assert deform_conv_backward_parameters_cuda(cu(xn), cu(on), cu(gn), gwt, None, None, *args, 1.0, 2) == 1
    for got, ref in ((gi, gx), (go_, goff), (gwt, gw)):
        assert np.abs(got.cpu().numpy() - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
    # AlignConv geometry with channel counts the LDS-accumulating col2im takes (3x3, stride 1, C % 8 == 0),
    # offsets large enough that some samples leave the tile's window (global-atomic path)
    B, C, H, W, O = 2, 16, 19, 45, 8
    xn = rng.standard_normal((B, C, H, W)).astype(np.float32)
    wn = (rng.standard_normal((O, C, 3, 3)) * 0.2).astype(np.float32)
    on = (rng.standard_normal((B, 18, H, W)) * 4.0).astype(np.float32)
    gn = rng.standard_normal((B, O, H, W)).astype(np.float32)
    gx, goff, gw = oracle.deform_conv_backward(xn, on, wn, gn)
    gi, go_ = torch.zeros(B, C, H, W, device=dev()), torch.zeros(B, 18, H, W, device=dev())
    assert deform_conv_backward_input_cuda(cu(xn), cu(on), cu(gn), gi, go_, cu(wn), None, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 2) == 1
    for got, ref in ((gi, gx), (go_, goff)):
        assert np.abs(got.cpu().numpy() - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
    # f16 storage (columns and GEMMs in half as the reference's half path): loose tolerance
    xh, oh, wh = (cu(g[k]).half().requires_grad_(True) for k in ("x", "offset", "weight"))
    deform_conv(xh, oh, wh, 1, 1, 1, 1, 1).backward(cu(g["grad_out"]).half())
    for got, ref in ((xh.grad, g["grad_input"]), (oh.grad, g["grad_offset"]), (wh.grad, g["grad_weight"])):
        assert np.abs(got.float().cpu().numpy() - ref).max() < 3e-2 * max(1.0, np.abs(ref).max())


def test_dcn_backward_input_fused_f16_vs_oracle(rng, monkeypatch):
    """the fused f16 input / offset gradient (s2a_deform_conv_backward_input_f16: column gradient on the matrix cores,
    consumed in LDS, no `columns`) against the oracle (deform_conv_cuda.cpp:262-374 restated in f32) on the f16-rounded
    operands, and against the unfused path (library GEMM + col2im kernels) on the same call: tame offsets (everything
    inside the LDS window), wild ones (global-atomic path), a ragged image (partly filled tiles), two channel chunks"""
    from s2anet_amd.dcn import deform_conv_backward_input_cuda
    # (the last two: H W % 8 == 0 -- the 16-bytes-per-lane staging copy, with a part-filled channel tile and a part-filled position tile)
    for (B, C, H, W, O, amp) in ((2, 64, 19, 45, 32, 0.7), (1, 32, 9, 20, 16, 5.0), (2, 64, 8, 16, 48, 2.0), (1, 96, 5, 8, 16, 1.0)):
        xn = rng.standard_normal((B, C, H, W)).astype(np.float16)
        wn = (rng.standard_normal((O, C, 3, 3)) * 0.1).astype(np.float16)
        on = (rng.standard_normal((B, 18, H, W)) * amp).astype(np.float16)
        gn = rng.standard_normal((B, O, H, W)).astype(np.float16)
        gx, goff, _ = oracle.deform_conv_backward(xn.astype(np.float32), on.astype(np.float32), wn.astype(np.float32),
                                                  gn.astype(np.float32))
        outs = {}
        for mode in ("fused", "unfused"):
            if mode == "unfused":
                monkeypatch.setenv("S2A_DCN_BWD_UNFUSED", "1")
            else:
                monkeypatch.delenv("S2A_DCN_BWD_UNFUSED", raising=False)
            gi = torch.zeros(B, C, H, W, device=dev(), dtype=torch.float16)
            go_ = torch.zeros(B, 18, H, W, device=dev(), dtype=torch.float16)
            assert deform_conv_backward_input_cuda(cu(xn), cu(on), cu(gn), gi, go_, cu(wn), None, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, B) == 1
            outs[mode] = (gi.float().cpu().numpy(), go_.float().cpu().numpy())
        for mode, (gi, go_) in outs.items():
            # f16 outputs; the fused path accumulates in f32 and rounds once (the unfused one keeps f16 columns)
            tol_i = (4e-3 if mode == "fused" else 2e-2) * max(1.0, np.abs(gx).max())
            tol_o = (4e-3 if mode == "fused" else 3e-2) * max(1.0, np.abs(goff).max())
            assert np.abs(gi - gx).max() < tol_i, (mode, (B, C, H, W, O), np.abs(gi - gx).max(), np.abs(gx).max())
            assert np.abs(go_ - goff).max() < tol_o, (mode, (B, C, H, W, O), np.abs(go_ - goff).max(), np.abs(goff).max())
        # the C entry that ACCUMULATES into the caller's f32 tensor (the Python side goes through the typed entry since round 6):
        # += on a non-zero tensor, same tolerance
        from s2anet_amd import _lib
        L = _lib.lib()
        base = rng.standard_normal((B, C, H, W)).astype(np.float32)
        acc, goff16 = cu(base).clone(), torch.empty(B, 18, H, W, device=dev(), dtype=torch.float16)
        ws = _lib.workspace(L.s2a_deform_conv_backward_input_workspace_bytes(B, C, H, W, O), dev(), "dcn_bwd")
        xt, ot, gt, wt = cu(xn), cu(on), cu(gn), cu(wn)
        _lib.check(L.s2a_deform_conv_backward_input_f16(_lib.ptr(xt), _lib.ptr(ot), _lib.ptr(gt), _lib.ptr(wt), _lib.ptr(acc),
                                                        _lib.ptr(goff16), B, C, H, W, O, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev())))
        assert np.abs(acc.cpu().numpy() - base - gx).max() < 4e-3 * max(1.0, np.abs(gx).max())
        assert np.abs(goff16.float().cpu().numpy() - goff).max() < 4e-3 * max(1.0, np.abs(goff).max())
        assert np.abs(gx).max() > 0.5 and np.abs(goff).max() > 0.5


def test_dcn_backward_input_fused_f32_vs_oracle(rng, monkeypatch):
    """the fused f32 input / offset gradient (s2a_deform_conv_backward_input_f32: column gradient on the f32 matrix
    instruction, 4 x 8 tiles, accumulated straight into the caller's gradInput) against the oracle
    (deform_conv_cuda.cpp:262-374 restated) within the north_star's 1e-4, and against the unfused path on the same call:
    tame offsets, wild ones (global-atomic path), ragged images, one / two / three channel chunks, O = 16 ... 256,
    a non-zero gradInput (accumulation), a non-contiguous gradInput (staged)"""
    from s2anet_amd.dcn import deform_conv_backward_input_cuda
    cases = ((2, 64, 19, 45, 32, 0.7), (1, 32, 9, 20, 16, 5.0), (2, 96, 8, 16, 48, 2.0), (1, 32, 13, 11, 256, 1.0))
    for (B, C, H, W, O, amp) in cases:
        xn = rng.standard_normal((B, C, H, W)).astype(np.float32)
        wn = (rng.standard_normal((O, C, 3, 3)) * 0.1).astype(np.float32)
        on = (rng.standard_normal((B, 18, H, W)) * amp).astype(np.float32)
        gn = rng.standard_normal((B, O, H, W)).astype(np.float32)
        base = rng.standard_normal((B, C, H, W)).astype(np.float32)
        gx, goff, _ = oracle.deform_conv_backward(xn, on, wn, gn)
        for mode in ("fused", "unfused", "fused-staged"):
            if mode == "unfused":
                monkeypatch.setenv("S2A_DCN_BWD_UNFUSED", "1")
            else:
                monkeypatch.delenv("S2A_DCN_BWD_UNFUSED", raising=False)
            gi = cu(base).clone()
            if mode == "fused-staged":
                gi = torch.empty(B, C, W, H, device=dev()).transpose(2, 3)
                gi.copy_(cu(base))
                assert not gi.is_contiguous()
            go_ = torch.full((B, 18, H, W), 7.0, device=dev())
            assert deform_conv_backward_input_cuda(cu(xn), cu(on), cu(gn), gi, go_, cu(wn), None, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, B) == 1
            ei = np.abs(gi.cpu().numpy() - base - gx).max()
            eo = np.abs(go_.cpu().numpy() - goff).max()
            assert ei < 1e-4 * max(1.0, np.abs(gx).max()), (mode, (B, C, H, W, O), ei, np.abs(gx).max())
            assert eo < 1e-4 * max(1.0, np.abs(goff).max()), (mode, (B, C, H, W, O), eo, np.abs(goff).max())
        assert np.abs(gx).max() > 0.5 and np.abs(goff).max() > 0.5


def test_dcn_backward_weight_fused_f16_vs_oracle(rng, monkeypatch):
    """the fused f16 weight gradient (s2a_deform_conv_backward_weight_f16: columns formed in LDS, contracted over the
    positions on the matrix cores through transposing LDS reads, per-workgroup partial blocks summed by
    k_dcn_bwd_weight_reduce in a fixed order -- no atomics) against the oracle
    (deform_conv_cuda.cpp:376-489 restated) on the f16-rounded operands and against the unfused path (im2col + library
    GEMM): tame and wild offsets, ragged images, one and two channel chunks, fewer out channels than waves, scale"""
    from s2anet_amd.dcn import deform_conv_backward_parameters_cuda
    for (B, C, H, W, O, amp, scale) in ((2, 64, 19, 45, 32, 0.7, 1.0), (1, 128, 9, 20, 64, 5.0, 0.5), (3, 64, 8, 16, 256, 2.0, 1.0)):
        xn = rng.standard_normal((B, C, H, W)).astype(np.float16)
        on = (rng.standard_normal((B, 18, H, W)) * amp).astype(np.float16)
        gn = (rng.standard_normal((B, O, H, W)) * 0.5).astype(np.float16)
        w0 = np.zeros((O, C, 3, 3), np.float32)
        _, _, gw = oracle.deform_conv_backward(xn.astype(np.float32), on.astype(np.float32), w0, gn.astype(np.float32))
        gw = gw * scale
        for mode in ("fused", "unfused"):
            if mode == "unfused":
                monkeypatch.setenv("S2A_DCN_BWD_UNFUSED", "1")
            else:
                monkeypatch.delenv("S2A_DCN_BWD_UNFUSED", raising=False)
            gwt = torch.zeros(O, C, 3, 3, device=dev(), dtype=torch.float32)
            assert deform_conv_backward_parameters_cuda(cu(xn), cu(on), cu(gn), gwt, None, None, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1,
                                                        scale, B) == 1
            err = np.abs(gwt.cpu().numpy() - gw).max()
            # columns are rounded to f16 either way; the sums run over B * H * W positions in f32
            assert err < 6e-3 * max(1.0, np.abs(gw).max()), (mode, (B, C, H, W, O), err, np.abs(gw).max())
        assert np.abs(gw).max() > 1.0


def test_dcn_backward_weight_fused_f32_vs_oracle(rng, monkeypatch):
    """the fused f32 weight gradient (s2a_deform_conv_backward_weight_f32: columns formed in LDS, contracted over the
    positions on the matrix cores, partial blocks + deterministic reduce into the caller's gradWeight, scaled) against the oracle
    (deform_conv_cuda.cpp:376-489 restated) within the north_star's 1e-4 and against the unfused path: tame and wild
    offsets, ragged images, one and two channel chunks, fewer out channels than waves, scale, a non-zero gradWeight.
    Both fused kernels: the default (round 6: the f32 tensors as three bf16 planes, six 16-bit products per f32 product --
    k_dcn_bwd_weight_x3) and the f32 matrix instruction's (S2A_BWD_F32_WEIGHT=mfma32); the two agree to ~1e-6 of the largest
    entry, two orders inside the bound, and each is bit-identical from call to call"""
    from s2anet_amd.dcn import deform_conv_backward_parameters_cuda
    for (B, C, H, W, O, amp, scale) in ((2, 64, 19, 45, 32, 0.7, 1.0), (1, 128, 9, 20, 64, 5.0, 0.5), (3, 64, 8, 16, 256, 2.0, 1.0)):
        xn = rng.standard_normal((B, C, H, W)).astype(np.float32)
        on = (rng.standard_normal((B, 18, H, W)) * amp).astype(np.float32)
        gn = (rng.standard_normal((B, O, H, W)) * 0.5).astype(np.float32)
        if O == 256:
            gn[0, :, 2, 3] *= 1e-3          # magnitudes spread over many binades: the split must carry small and large values alike
            xn[0, ::7] *= 300.0
        base = rng.standard_normal((O, C, 3, 3)).astype(np.float32)
        _, _, gw = oracle.deform_conv_backward(xn, on, np.zeros((O, C, 3, 3), np.float32), gn)
        gw = gw * scale
        got = {}
        for mode in ("fused", "fused-again", "fused-mfma32", "unfused"):
            monkeypatch.delenv("S2A_DCN_BWD_UNFUSED", raising=False)
            monkeypatch.delenv("S2A_BWD_F32_WEIGHT", raising=False)
            if mode == "unfused":
                monkeypatch.setenv("S2A_DCN_BWD_UNFUSED", "1")
            if mode == "fused-mfma32":
                monkeypatch.setenv("S2A_BWD_F32_WEIGHT", "mfma32")
            gwt = cu(base).clone()
            assert deform_conv_backward_parameters_cuda(cu(xn), cu(on), cu(gn), gwt, None, None, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1,
                                                        scale, B) == 1
            got[mode] = gwt.cpu().numpy() - base
            err = np.abs(got[mode] - gw).max()
            assert err < 1e-4 * max(1.0, np.abs(gw).max()), (mode, (B, C, H, W, O), err, np.abs(gw).max())
        assert np.array_equal(got["fused"], got["fused-again"])
        d = np.abs(got["fused"] - got["fused-mfma32"]).max()
        assert d < 1e-5 * max(1.0, np.abs(gw).max()), (d, np.abs(gw).max())
        assert np.abs(gw).max() > 1.0


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_dcn_backward_autograd_one_call(rng, monkeypatch, dtype):
    """DeformConvFunction.backward through s2a_deform_conv_backward (both gradients in one library call, the NHWC copies of
    input and gradOutput shared) == the two entry points called one after the other (deform_conv.py:73-118), and == the
    oracle; the weight gradient is reduced in a fixed order, so it is bit-identical between the two and from run to run"""
    from s2anet_amd.dcn import deform_conv
    td = getattr(torch, dtype)
    B, C, H, W, O = 2, 64, 13, 21, 32
    xn = rng.standard_normal((B, C, H, W)).astype(np.float32)
    wn = (rng.standard_normal((O, C, 3, 3)) * 0.1).astype(np.float32)
    on = (rng.standard_normal((B, 18, H, W)) * 1.5).astype(np.float32)
    gn = rng.standard_normal((B, O, H, W)).astype(np.float32)
    if dtype == "float16":
        xn, wn, on, gn = (a.astype(np.float16).astype(np.float32) for a in (xn, wn, on, gn))
    gx, goff, gw = oracle.deform_conv_backward(xn, on, wn, gn)
    res = {}
    for mode in ("one-call", "one-call-again", "separate"):
        if mode == "separate":
            monkeypatch.setenv("S2A_DCN_BWD_SEPARATE", "1")
        else:
            monkeypatch.delenv("S2A_DCN_BWD_SEPARATE", raising=False)
        x, off, w = (cu(a).to(td).requires_grad_(True) for a in (xn, on, wn))
        deform_conv(x, off, w, 1, 1, 1, 1, 1).backward(cu(gn).to(td))
        res[mode] = tuple(t.grad.float().cpu().numpy() for t in (x, off, w))
        assert x.grad.dtype == td and off.grad.dtype == td and w.grad.dtype == td
    tol = 1e-4 if dtype == "float32" else 6e-3
    for mode, (a, b, c) in res.items():
        for got, ref in ((a, gx), (b, goff), (c, gw)):
            assert np.abs(got - ref).max() < tol * max(1.0, np.abs(ref).max()), (mode, np.abs(got - ref).max(), np.abs(ref).max())
    assert np.array_equal(res["one-call"][2], res["one-call-again"][2]) and np.array_equal(res["one-call"][2], res["separate"][2])
    assert np.array_equal(res["one-call"][1], res["separate"][1])


@pytest.mark.parametrize("form", ["list", "matrix"])
def test_assign_labels_fused(rng, monkeypatch, form):
    """fused label assignment == the reference's assign_labels (golden from its own Python on its CPU IoU op;
    the two sort branches of the IoU agree on these inputs) and == the oracle on fresh inputs.  Both forms: the list form
    (round 6: no IoU matrix, the default up to 1 024 gts / 8 Mi pairs) and the matrix form (S2A_ASSIGN_LIST=0, also taken beyond)"""
    from s2anet_amd.rotated import assign_labels
    from test_oracle_pinned import ASSIGN_CASES
    monkeypatch.setenv("S2A_ASSIGN_LIST", "1" if form == "list" else "0")
    g = golden("assign_labels.npz")
    for tag, kw in ASSIGN_CASES:
        got = assign_labels(cu(g["anchors"]), cu(g["gts"]), **kw).cpu().numpy()
        assert np.array_equal(got, oracle.assign_labels(g["anchors"], g["gts"], **kw)), tag
        assert np.array_equal(got, g["assign_" + tag]), tag
    assert np.array_equal(assign_labels(cu(g["anchors"]), torch.zeros((0, 5), device=dev())).cpu().numpy(), g["assign_empty"])
    # the real shape: all 21 824 grid anchors of a 1024^2 chip against 300 ground-truth boxes
    a = np.concatenate([oracle.grid_anchors(1024 // s, 1024 // s, s).reshape(-1, 5) for s in (8, 16, 32, 64, 128)]).astype(np.float32)
    a[:, 4] = rng.uniform(-0.7, 2.3, a.shape[0])
    gt = rand_rboxes(rng, 300, span=1024)
    gt[:50] = a[rng.choice(a.shape[0], 50, replace=False)] * np.array([1, 1, 1.1, 0.9, 1], np.float32)
    got = assign_labels(cu(a), cu(gt)).cpu().numpy()
    assert np.array_equal(got, oracle.assign_labels(a, gt))
    assert (got >= 0).sum() >= 50
    # one anchor per gt, anchors outside the image, 32 gts, exact ties (two gts on the same anchor, a gt copied twice)
    gt32 = gt[:32].copy()
    gt32[5] = gt32[4]
    a2 = a.copy()
    a2[::97, 0] = -5.0                                              # invalid anchors (:63-69)
    for kw in (dict(gt_max_assign_all=False), dict(gt_max_assign_all=True, pos_iou_thr=0.3, neg_iou_thr=0.1, min_pos_iou_thr=0.2),
               dict(filter_invalid_anchors=False), dict(imgs_size=(800, 900))):
        got = assign_labels(cu(a2), cu(gt32), **kw).cpu().numpy()
        assert np.array_equal(got, oracle.assign_labels(a2, gt32, **kw)), kw
    # a dense pile: every anchor of a small patch overlaps every gt (the list holds all M x N pairs)
    ad = rand_rboxes(rng, 700, span=60, lo=40, hi=90)
    gd = rand_rboxes(rng, 200, span=60, lo=40, hi=90)
    for kw in (dict(), dict(gt_max_assign_all=False)):
        got = assign_labels(cu(ad), cu(gd), imgs_size=(2000, 2000), **kw).cpu().numpy()
        assert np.array_equal(got, oracle.assign_labels(ad, gd, imgs_size=(2000, 2000), **kw)), kw
    # more gts than the tile form holds: the matrix form answers whatever the switch says
    gbig = rand_rboxes(rng, 1100, span=1024)
    got = assign_labels(cu(a), cu(gbig)).cpu().numpy()
    assert np.array_equal(got, oracle.assign_labels(a, gbig))


def test_voc_eval_on_gpu(rng):
    """Task-1 evaluation with the overlap search on the GPU == the reference script's voc_eval (golden) and the oracle"""
    from s2anet_amd.evaluate import voc_eval_arrays, polyiou_match
    from test_oracle_pinned import VOC_CASES
    g = golden("voc_eval.npz")
    a = (g["det_polys"], g["det_scores"], g["det_image"], g["gt_polys"], g["gt_image"], g["gt_difficult"], int(g["num_images"]))
    for tag, kw in VOC_CASES:
        rec, prec, ap, _ = voc_eval_arrays(*a, device=dev(), **kw)
        assert np.array_equal(rec, g["rec_" + tag]) and np.array_equal(prec, g["prec_" + tag]), tag
        assert ap == float(g["ap_" + tag]), tag
    # per-detection overlaps bit-exact against the oracle on a larger random case
    n_img, D, G = 40, 3000, 600
    gt = oracle.rboxes_to_polys(rand_rboxes(rng, G, span=500)); gi = np.sort(rng.integers(0, n_img, G))
    dp = oracle.rboxes_to_polys(rand_rboxes(rng, D, span=500)); di = rng.integers(0, n_img, D)
    sc = (rng.permutation(D) + 1.0) / (D + 1.0)
    _, _, _, _, (ovs, args) = oracle.voc_eval_arrays(dp, sc, di, gt, gi, np.zeros(G), n_img)
    order = np.argsort(-sc)
    off = np.zeros(n_img + 1, np.int64); np.add.at(off, gi + 1, 1); off = np.cumsum(off)
    ov, am = polyiou_match(cu(dp[order]), torch.from_numpy(di[order].astype(np.int32)).to(dev()), cu(gt),
                           torch.from_numpy(off).to(dev()))
    assert np.array_equal(ov.cpu().numpy(), ovs) and np.array_equal(am.cpu().numpy(), args)


def test_chip_merge_file_matches_reference_script(tmp_path):
    """mergesingle (chip-name parsing, poly2origpoly, per-image polygon NMS, output formatting) with the NMS on the
    GPU produces the reference script's output file line for line"""
    from s2anet_amd.merge import merge_lines, mergesingle, parse_chip_name
    g = golden("merge_file.npz")
    assert parse_chip_name("P0003__0.5__824___1648") == ("P0003", 824, 1648, "0.5")
    lines = [str(x) for x in g["lines"]]
    assert merge_lines(lines, 0.5, dev()) == [str(x) for x in g["merged"]]
    src = tmp_path / "Task1_ship.txt"
    src.write_text("\n".join(lines) + "\n")
    (tmp_path / "out").mkdir()
    mergesingle(str(tmp_path / "out"), str(src), 0.5, dev())
    assert (tmp_path / "out" / "Task1_ship.txt").read_text().splitlines() == [str(x) for x in g["merged"]]


def test_output_formats(rng):
    """rbox -> polygon (boxPoints restated; OpenCV absent: checked against the oracle restatement and the geometry),
    scale_coords_rotated, Task-1 lines"""
    from s2anet_amd.formats import rbox_to_poly, scale_coords_rotated, task1_lines
    b = rand_rboxes(rng, 500, span=900)
    b[0] = [100, 50, 40, 20, 0.0]
    b[1] = [100, 50, 40, 20, np.pi / 2 - 1e-3]
    b[2] = [100, 50, 40, 20, -0.5]
    got = rbox_to_poly(cu(b)).cpu().numpy()
    ref = oracle.rbox_to_poly(b)
    assert np.abs(got - ref).max() < 1e-3
    # the polygon is the rectangle: same centre, edge lengths {w, h}, polyiou with the 4-corner form == 1
    assert np.abs(got.reshape(-1, 4, 2).mean(1) - b[:, :2]).max() < 1e-2
    assert np.abs(oracle.polyiou(got.astype(np.float64), oracle.rboxes_to_polys(b)) - 1).max() < 1e-4
    d = torch.tensor([[110.0, 60, 40, 20, 0.3, 0.9], [510.0, 300, 80, 10, 1.2, 0.5]], device=dev())
    s = scale_coords_rotated((1024, 1024), d.clone(), (2048, 1000))
    assert torch.allclose(s[:, :4].cpu(), torch.tensor([[(110 - 262) / 0.5, 120, 80, 40], [(510 - 262) / 0.5, 600, 160, 20]]))
    lines = task1_lines("P0001.png", d, torch.tensor([1, 0], device=dev()), ["plane", "ship"])
    assert set(lines) == {"plane", "ship"} and lines["ship"][0].startswith("P0001 0.9000 ") and lines["ship"][0].count(" ") == 9


# ---------------------------------------------------------------- BASELINE.json full sizes: size-independent properties
def test_full_size_iou_10k_x_10k(rng):
    """config 1: 10 000 x 10 000 rotated IoU -- range, sparsity, self-overlap, and 30 000 sampled entries bit-exact
    against the oracle (the oracle cannot do 10^8 pairs in test time)"""
    import s2anet_amd as S
    b1, b2 = rand_rboxes(rng, 10000, span=1024), rand_rboxes(rng, 10000, span=1024)
    iou = S.box_iou_rotated(cu(b1), cu(b2))
    assert iou.shape == (10000, 10000)
    assert iou.min().item() >= 0.0 and iou.max().item() <= 1.0 + 1e-5
    frac = (iou > 0).float().mean().item()
    assert 0.003 < frac < 0.03                                       # ~1.1 % of DOTA-like pairs overlap
    ii, jj = rng.integers(0, 10000, 30000), rng.integers(0, 10000, 30000)
    nz = (iou > 0).nonzero()[:15000].cpu().numpy()                   # plus 15 000 overlapping ones
    ii, jj = np.concatenate([ii, nz[:, 0]]), np.concatenate([jj, nz[:, 1]])
    ref = oracle.iou_pairs(b1[ii], b2[jj], sort_mode=oracle.SORT_GPU)
    got = iou[torch.from_numpy(ii).to(dev()), torch.from_numpy(jj).to(dev())].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    from s2anet_amd.rotated import box_iou_rotated_pairs
    self_iou = box_iou_rotated_pairs(cu(b1[:4000]), cu(b1[:4000])).cpu().numpy()
    assert np.abs(self_iou - 1).max() < 1e-4


def test_full_size_ml_nms_200k(rng):
    """config 5: 200 000 rows x 15 labels -- descending order, idempotence, and the keep list of three whole label
    slices (~13 k rows each) of the SAME 200 k-row call against the oracle (`>` rule, GPU sort branch): the Morton /
    spatial order, k_nms_cull_lanes (>= 49 152 rows) and the tile list at full size meet the oracle, not themselves.
    ml-NMS decomposes by label exactly (utils/ml_nms_rotated/src/nms_rotated_cuda.cu:54-56: other-label pairs never
    suppress), so the rows of label c kept by the big call must be the oracle's NMS of that slice."""
    import s2anet_amd as S
    n = 200000
    d, s = rand_rboxes(rng, n, span=1024), distinct_scores(rng, n)
    lab = rng.integers(0, 15, n).astype(np.float32)
    D, Sc, Lb = cu(d), cu(s), cu(lab)
    k = S.ml_nms_rotated(D, Sc, Lb, 0.5)
    assert 100000 < k.numel() < n
    ks = Sc[k]
    assert (ks[1:] < ks[:-1]).all()
    k2 = S.ml_nms_rotated(D[k], Sc[k], Lb[k], 0.5)
    assert k2.numel() == k.numel() and (k2 == torch.arange(k.numel(), device=k.device)).all()
    kept = np.zeros(n, bool)
    kh = k.cpu().numpy()
    kept[kh] = True
    for c in (0, 7, 14):
        idx = np.nonzero(lab == c)[0]
        assert 12000 < idx.size < 15000
        ko = oracle.nms_rotated(d[idx], s[idx], 0.5, rule=oracle.RULE_GT, sort_mode=oracle.SORT_GPU, cull=True)
        assert np.array_equal(np.sort(idx[ko]), idx[kept[idx]]), c
        # and in the call's own order: the kept rows of label c appear in descending score, as the oracle lists them
        assert np.array_equal(kh[lab[kh] == c], idx[ko]), c


def test_full_size_alignconv_zero_offset_identity():
    """config 2 at batch 8: anchors that sit on the sampling grid (centre = index * stride, side 3 * stride, angle 0)
    give zero offsets, so AlignConv == 3x3 convolution + ReLU (SURVEY 8(c) ii) at the full P3 size, f16.
    A size-independent PROPERTY (two kernels of this library against each other), not a parity test: configs[1] / [2]
    parity against the oracle is test_config1_* / test_config2_* in test_gpu_e2e.py"""
    from s2anet_amd.alignconv import align_conv_forward
    from s2anet_amd.fused import conv_f16, conv_pack_weight
    B, C, H, W, O, st = 8, 256, 128, 128, 256, 8
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, C, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(O, C, 3, 3, generator=g) * 0.02).to(dev()).half()
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    anc = torch.zeros(B, H, W, 5)
    anc[..., 0], anc[..., 1], anc[..., 2], anc[..., 3] = xs * st, ys * st, 3.0 * st, 3.0 * st
    out = align_conv_forward(x, anc.to(dev()), w, st, relu=True)
    ref = conv_f16(x, conv_pack_weight(w), None, O, 3, 1, True)
    assert out.shape == ref.shape
    d = (out.float() - ref.float()).abs()
    assert d.max().item() < 2e-2 and (d > 0).float().mean().item() < 0.05


def test_head_on_concurrent_streams_matches_serial():
    """bench.py keeps three independent batches in flight on three HIP streams: every per-call scratch buffer must
    be private to its stream.  The whole head + post-processing (own kernels only; the library convolutions of the
    trunk are not run-to-run deterministic) on three different feature pyramids issued back to back on three
    streams == the same inputs run one at a time, bit for bit."""
    from s2anet_amd.detector import build_synthetic_detector
    from s2anet_amd.pyramid import PyramidLayout
    m = build_synthetic_detector(device=dev())
    m.head.odm_cls_head.bias.data.fill_(-2.0)
    m.head.odm_cls_head.weight.data.mul_(20.0)                 # spread the scores: thousands of candidates, few ties
    layout = PyramidLayout(2, [(40, 48), (20, 24), (10, 12), (5, 6), (3, 3)], (8, 16, 32, 64, 128))   # no level > 2000 positions
    g = torch.Generator().manual_seed(77)
    feats = [torch.randn(layout.pixels, 256, generator=g).to(dev()).half() for _ in range(3)]

    def run(x):
        return m.head.get_bboxes_batched(m.head.forward_pyramid(layout, x))
    with torch.no_grad():
        ref = [tuple(t.clone() for t in run(x)) for x in feats]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=dev()) for _ in range(3)]
        for rep in range(3):                                       # warm per-stream pools, then the checked round
            outs = []
            for x, st in zip(feats, streams):
                with torch.cuda.stream(st):
                    outs.append(run(x))
            torch.cuda.synchronize()
    assert int(ref[0][2].sum()) > 100
    for (d, l, c), (rd, rl, rc) in zip(outs, ref):
        assert torch.equal(c, rc) and torch.equal(l, rl) and torch.equal(d, rd)


def test_pyramid_topk_ties_and_sizes():
    """the per-(image, level) select on heavily tied keys (a handful of distinct logits): exactly k rows, every row
    above the k-th key, the k-th key's rows in ascending position order, positions ascending; levels at, below and
    above k; 17+ classes take the generic key loop"""
    from s2anet_amd import pyramid as P
    from s2anet_amd.pyramid import PyramidLayout
    g = torch.Generator().manual_seed(31)
    for C, sizes, k in ((15, [(96, 96), (45, 45), (44, 45), (5, 7)], 2000), (20, [(70, 90), (10, 10)], 333), (3, [(150, 160)], 2000)):
        layout = PyramidLayout(2, sizes, tuple(8 * 2 ** i for i in range(len(sizes))))
        cls = torch.full((layout.pixels, 64), 9.0)                   # columns >= C must be ignored
        cls[:, :C] = (torch.randint(-3, 4, (layout.pixels, C), generator=g).float() * 0.25)
        cls = cls.half().to(dev())
        reg = torch.zeros(layout.pixels, 64).half().to(dev())
        anc = (torch.rand(layout.pixels, 5, generator=g) * 50 + 8).to(dev())
        out = P.candidates(layout, cls, reg, anc, C, k)
        assert out is not None
        sel = out[2].cpu().numpy()
        key = cls[:, :C].float().max(1)[0].cpu().numpy()
        off = 0
        for l, (h, w) in enumerate(sizes):
            hw, m = h * w, min(h * w, k)
            for b in range(layout.batch):
                r0 = layout.pix0[l] + b * hw
                got = sel[b, off:off + m]
                assert (np.diff(got) > 0).all() and got.min() >= r0 and got.max() < r0 + hw, (C, l, b)
                kl = key[r0:r0 + hw]
                if hw <= k:
                    assert np.array_equal(got, np.arange(r0, r0 + hw))
                    continue
                T = np.sort(kl)[::-1][k - 1]
                above = np.nonzero(kl > T)[0]
                equal = np.nonzero(kl == T)[0][:k - len(above)]          # lowest positions first
                assert np.array_equal(got, np.sort(np.concatenate([above, equal])) + r0), (C, l, b)
            off += m


def test_pyramid_candidates_vs_stock_path():
    """fused per-level top-k + sigmoid + decode (two kernels) against the stock-op path of S2ANetHead.candidates:
    same candidate set per level (distinct scores -> no tie at the k-th place), same boxes and scores"""
    from s2anet_amd import pyramid as P
    from s2anet_amd.head import S2ANetHead
    from s2anet_amd.pyramid import PyramidLayout
    layout = PyramidLayout(2, [(64, 80), (32, 40), (16, 20), (8, 10), (4, 5)], (8, 16, 32, 64, 128))   # 5120 / 1280 / ... positions
    g = torch.Generator().manual_seed(9)
    C = 15
    cls = torch.zeros(layout.pixels, 64)
    # distinct max logits per position: a random permutation of a fine grid, exactly representable in f16
    base = (torch.randperm(layout.pixels, generator=g).float() - layout.pixels / 2) / 2048.0
    cls[:, :C] = base[:, None] - torch.rand(layout.pixels, C, generator=g) * 0.5
    cls[torch.arange(layout.pixels), torch.randint(0, C, (layout.pixels,), generator=g)] = base
    cls = cls.half().to(dev())
    reg = torch.zeros(layout.pixels, 64)
    reg[:, :5] = torch.randn(layout.pixels, 5, generator=g) * 0.3
    reg = reg.half().to(dev())
    anc = torch.rand(layout.pixels, 5, generator=g) * torch.tensor([1000.0, 1000, 80, 80, 2.0]) + torch.tensor([0.0, 0, 8, 8, -0.7])
    anc = anc.to(dev())
    head = S2ANetHead(C).to(dev())
    k = head.max_before_nms_per_level
    bb, sc, sel = P.candidates(layout, cls, reg, anc, C, k)
    n = len(layout.sizes)
    p = (None, None, [layout.level(cls, l, C) for l in range(n)], [layout.level(reg, l, 5) for l in range(n)],
         [layout.rows(anc, l).view(layout.batch, *layout.sizes[l], 5) for l in range(n)])
    with torch.no_grad():
        rb, rs = head.candidates(p)
    assert bb.shape == rb.shape and sc.shape == rs.shape
    # the stock path orders a top-k level by score, the fused one by position: compare as sets per (image, level)
    off = 0
    for l, (h, w) in enumerate(layout.sizes):
        m = min(h * w, k)
        for b in range(layout.batch):
            a = torch.cat([bb[b, off:off + m], sc[b, off:off + m]], 1)
            r = torch.cat([rb[b, off:off + m], rs[b, off:off + m]], 1)
            ka = a[:, 5:].max(1)[0].argsort()
            kr = r[:, 5:].max(1)[0].argsort()
            assert torch.allclose(a[ka], r[kr], rtol=1e-5, atol=1e-5), (l, b)
        off += m
    # positions come in ascending order inside a level and stay inside the image's rows of that level
    s0 = sel[:, :min(64 * 80, k)]
    assert (s0[:, 1:] > s0[:, :-1]).all()


def _offsets_away_from_integers(rng, shape, amp=1.6):
    """integer part anything, fractional part in [0.2, 0.8]: no sampling point on a grid line (where the bilinear sampler's
    derivative jumps) or within 0.2 of the image-border rule h_im > -1 / < H"""
    return np.floor(rng.uniform(-amp, amp, shape)) + rng.uniform(0.2, 0.8, shape)


def test_deform_conv_float64_forward_vs_oracle(rng):
    """the float64 instantiation (AT_DISPATCH_FLOATING_TYPES_AND_HALF, deform_conv_cuda_kernel.cu:258): generic geometry,
    groups and deformable groups, against the float64 restatement to 1e-12"""
    from s2anet_amd.dcn import deform_conv
    for (B, C, H, W, O, k, st, pd, dl, g, dg) in ((2, 4, 7, 9, 6, 3, 1, 1, 1, 1, 1), (1, 8, 9, 11, 6, 3, 2, 1, 2, 2, 2),
                                                  (2, 64, 12, 16, 64, 3, 1, 1, 1, 1, 1), (1, 6, 5, 6, 4, 1, 1, 0, 1, 1, 3)):
        Ho = (H + 2 * pd - (dl * (k - 1) + 1)) // st + 1
        Wo = (W + 2 * pd - (dl * (k - 1) + 1)) // st + 1
        x = rng.standard_normal((B, C, H, W))
        off = rng.standard_normal((B, dg * 2 * k * k, Ho, Wo)) * 1.5
        w = rng.standard_normal((O, C // g, k, k)) * 0.3
        got = deform_conv(cu(x), cu(off), cu(w), st, pd, dl, g, dg)
        assert got.dtype == torch.float64
        ref = oracle.deform_conv_forward_f64(x, off, w, (st, st), (pd, pd), (dl, dl), g, dg)
        assert np.abs(got.cpu().numpy() - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
    # a float64 tensor on an entry point without a float64 instantiation is refused, not silently narrowed
    from s2anet_amd.alignconv import align_conv_forward
    with pytest.raises(TypeError):
        align_conv_forward(cu(rng.standard_normal((1, 64, 8, 8))), cu(np.zeros((1, 8, 8, 5), np.float32)),
                           cu(rng.standard_normal((64, 64, 3, 3))), 8.0)


def test_deform_conv_float64_gradcheck(rng):
    """torch.autograd.gradcheck on the float64 instantiation ties the backward (deformable_col2im / col2im_coord / im2col +
    the library GEMMs, deform_conv_cuda.cpp:262-489) to the forward independently of the oracle -- as the reference's own
    smoke block does for ARF (models/orn/functions/active_rotating_filter.py:99).  [2,4,7,9], offsets away from integer
    coordinates; plus a strided / dilated / grouped geometry."""
    from s2anet_amd.dcn import deform_conv
    x = cu(rng.standard_normal((2, 4, 7, 9))).requires_grad_()
    off = cu(_offsets_away_from_integers(rng, (2, 18, 7, 9))).requires_grad_()
    w = cu(rng.standard_normal((3, 4, 3, 3)) * 0.5).requires_grad_()
    assert torch.autograd.gradcheck(lambda a, b, c: deform_conv(a, b, c, 1, 1, 1, 1, 1), (x, off, w), eps=1e-6, atol=1e-6,
                                    rtol=1e-5, nondet_tol=1e-9)
    x = cu(rng.standard_normal((1, 4, 8, 9))).requires_grad_()
    off = cu(_offsets_away_from_integers(rng, (1, 36, 3, 4), amp=1.2)).requires_grad_()
    w = cu(rng.standard_normal((4, 2, 3, 3)) * 0.5).requires_grad_()
    assert torch.autograd.gradcheck(lambda a, b, c: deform_conv(a, b, c, 2, 1, 2, 2, 2), (x, off, w), eps=1e-6, atol=1e-6,
                                    rtol=1e-5, nondet_tol=1e-9)


def test_deform_conv_float64_backward_vs_f32_oracle(rng):
    """the same float64 backward against the oracle's float32 restatement of deform_conv_cuda.cpp:262-489 (AlignConv
    geometry): float noise apart"""
    from s2anet_amd.dcn import deform_conv
    B, C, H, W, O = 2, 8, 9, 10, 6
    xn, wn = rng.standard_normal((B, C, H, W)).astype(np.float32), (rng.standard_normal((O, C, 3, 3)) * 0.2).astype(np.float32)
    on = _offsets_away_from_integers(rng, (B, 18, H, W)).astype(np.float32)
    gn = rng.standard_normal((B, O, H, W)).astype(np.float32)
    gx, goff, gw = oracle.deform_conv_backward(xn, on, wn, gn)
    x, off, w = (cu(a.astype(np.float64)).requires_grad_() for a in (xn, on, wn))
    deform_conv(x, off, w, 1, 1, 1, 1, 1).backward(cu(gn.astype(np.float64)))
    for got, ref in ((x.grad, gx), (off.grad, goff), (w.grad, gw)):
        assert got.dtype == torch.float64
        assert np.abs(got.cpu().numpy() - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())


def test_nms_deferred_score_order_and_its_fallback(rng, monkeypatch):
    """round 6: a synchronous ml_nms_rotated call on the spatial path no longer builds the score order A beside the cull (the
    overflow fallback alone reads it): the host checks the overflow word behind its synchronisation and builds it then.
    (a) ordinary input: same keep list with the order built up front (S2A_NMS_DEFER_A=0) and deferred; (b) more distinct labels
    than the order-B table numbers (status bit 2): the deferred fallback runs and gives the oracle's list; (c) a pile in which
    nearly every pair of a label overlaps (lists sized for it still overflow nothing -> exact path) -- all == oracle."""
    import s2anet_amd as S
    n = 20000
    d, s = rand_rboxes(rng, n), distinct_scores(rng, n)
    lab = rng.integers(0, 15, n).astype(np.float32)
    want = oracle.ml_nms_rotated(d, s, lab, 0.5)
    for mode in ("0", "1"):
        monkeypatch.setenv("S2A_NMS_DEFER_A", mode)
        assert np.array_equal(S.ml_nms_rotated(cu(d), cu(s), cu(lab), 0.5).cpu().numpy(), want), mode
    lab = np.arange(n, dtype=np.float32) * 0.25                           # 20 000 distinct labels: the table reports itself full
    want = oracle.ml_nms_rotated(d, s, lab, 0.5)
    assert len(want) == n
    for mode in ("0", "1"):
        monkeypatch.setenv("S2A_NMS_DEFER_A", mode)
        assert np.array_equal(S.ml_nms_rotated(cu(d), cu(s), cu(lab), 0.5).cpu().numpy(), want), mode
    monkeypatch.delenv("S2A_NMS_DEFER_A")
    n = 6000
    d = rand_rboxes(rng, n, span=80, lo=30, hi=90)
    s = distinct_scores(rng, n)
    lab = rng.integers(0, 2, n).astype(np.float32)
    assert np.array_equal(S.ml_nms_rotated(cu(d), cu(s), cu(lab), 0.3).cpu().numpy(), oracle.ml_nms_rotated(d, s, lab, 0.3))
