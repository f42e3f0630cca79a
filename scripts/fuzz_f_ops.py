"""randomised s2a_assign_labels and s2a_nms_poly calls against the oracle (both rewritten in round 6: list forms).  A bounded,
seeded slice runs in the test suite (tests/test_gpu_fuzz.py); more cases by hand: python scripts/fuzz_f_ops.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def rboxes(rng, n, span, lo, hi):
    b = np.empty((n, 5), np.float32)
    b[:, :2] = rng.uniform(0, span, (n, 2))
    b[:, 2:4] = rng.uniform(lo, hi, (n, 2))
    b[:, 4] = rng.uniform(-np.pi / 4, 3 * np.pi / 4, n)
    return b


def assign_case(rng):
    """one random assign_labels call -> (ok, description)"""
    import oracle
    from s2anet_amd.rotated import assign_labels
    M = int(rng.choice([1, 15, 16, 17, 300, 2000, 9000]))
    N = int(rng.choice([1, 2, 31, 64, 300, 1024, 1025]))
    span = float(rng.choice([80.0, 400.0, 1500.0]))
    hi = float(rng.choice([20.0, 120.0, 600.0]))
    a, g = rboxes(rng, M, span, 4, hi), rboxes(rng, N, span, 4, hi)
    if rng.random() < 0.4 and M > 4 and N > 2:           # exact ties: copies of anchors as gts, a gt twice
        k = min(N, M) // 2
        g[:k] = a[rng.choice(M, k, replace=False)]
        g[-1] = g[0]
    if rng.random() < 0.3:
        a[rng.integers(0, M, max(1, M // 10)), 0] = -3.0     # invalid anchors
    kw = dict(imgs_size=(int(span), int(span + 100)), gt_max_assign_all=bool(rng.integers(0, 2)),
              filter_invalid_anchors=bool(rng.integers(0, 2)), filter_invalid_ious=bool(rng.integers(0, 2)))
    if rng.random() < 0.5:
        kw.update(pos_iou_thr=0.3, neg_iou_thr=0.1, min_pos_iou_thr=float(rng.choice([0.0, 0.2])))
    got = assign_labels(torch.from_numpy(a).cuda(), torch.from_numpy(g).cuda(), **kw).cpu().numpy()
    want = oracle.assign_labels(a, g, **kw)
    return np.array_equal(got, want), "M %5d N %4d span %5.0f hi %4.0f %s: %d positives" % (M, N, span, hi, kw, int((want >= 0).sum()))


def poly_case(rng):
    """one random nms_poly call -> (ok, description)"""
    import oracle
    from s2anet_amd.rotated import nms_poly
    n = int(rng.choice([1, 2, 63, 64, 65, 257, 1000, 1025, 4000]))
    span = float(rng.choice([40.0, 300.0, 2000.0]))
    polys = oracle.rboxes_to_polys(rboxes(rng, n, span, 4, float(rng.choice([30.0, 150.0]))))
    if rng.random() < 0.3 and n > 4:
        polys[: n // 4] = polys[0]                                       # a pile of identical polygons
    if rng.random() < 0.3:
        polys[rng.integers(0, n, max(1, n // 8))] = polys[rng.integers(0, n, max(1, n // 8)), ::-1][:, [1, 0, 3, 2, 5, 4, 7, 6]]   # reversed winding
    sc = (rng.permutation(n) + 1.0) / (n + 1.0)
    if rng.random() < 0.3 and n > 8:
        sc[: n // 3] = sc[0]                                             # score ties
    dets = np.concatenate([polys, sc[:, None]], 1)
    thr = float(rng.choice([0.0, 0.1, 0.3, 0.7, 0.95]))
    got = nms_poly(torch.from_numpy(dets).cuda(), thr).cpu().numpy()
    want = oracle.nms_poly(dets, thr)
    return np.array_equal(got, want), "n %5d span %5.0f thr %.2f keep %d" % (n, span, thr, len(want))


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
    bad = 0
    for c in range(cases):
        for fn in (assign_case, poly_case):
            ok, msg = fn(rng)
            bad += not ok
            print("case %2d %-12s %s  %s" % (c, fn.__name__, "ok" if ok else "MISMATCH", msg), flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
