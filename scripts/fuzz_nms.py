"""randomised ml_nms_rotated / batched NMS against the oracle (the order-B counting sort, the clean-up kernel's phases, the
wire-buffer emit).  A bounded, seeded slice runs in the test suite (tests/test_gpu_fuzz.py); more cases by hand:
python scripts/fuzz_nms.py [cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def boxes(rng, n, span, smin, smax):
    return np.concatenate([rng.uniform(0, span, (n, 2)), rng.uniform(smin, smax, (n, 2)), rng.uniform(-0.8, 2.4, (n, 1))], 1).astype(np.float32)


SIZES = (1, 7, 63, 64, 65, 500, 3000, 4097, 9000, 20000)


def nms_case(rng, sizes=SIZES):
    """one random ml_nms_rotated call -> (ok, description)"""
    import s2anet_amd as S, oracle
    n = int(rng.choice(sizes))
    K = int(rng.choice([1, 2, 15, 40]))
    span = float(rng.choice([60.0, 300.0, 1500.0])) * max(1.0, (n / 500) ** 0.5)
    d = boxes(rng, n, span, 4, float(rng.choice([20, 80])))
    if rng.random() < 0.3: d[: n // 3, :2] = d[0, :2]                      # a pile of boxes at one centre
    s = rng.permutation(n).astype(np.float32) / n + 0.001                  # distinct scores
    lab = rng.integers(0, K, n).astype(np.float32) * float(rng.choice([1.0, 0.5, -3.0]))
    thr = float(rng.choice([0.1, 0.3, 0.5]))
    got = S.ml_nms_rotated(torch.from_numpy(d).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(lab).cuda(), thr).cpu().numpy()
    want = oracle.ml_nms_rotated(d, s, lab, thr)
    return np.array_equal(got, want), "n %5d labels %2d span %6.0f thr %.1f keep %5d" % (n, K, span, thr, len(want))


def batched_case(rng, ns=(300, 2000, 5344)):
    """one detector-style batched call -> (ok, description)"""
    import s2anet_amd as S, oracle
    B, n, C = int(rng.choice([1, 3, 8])), int(rng.choice(ns)), 15
    bb = np.stack([boxes(rng, n, 1024, 8, 90) for _ in range(B)])
    sc = (rng.random((B, n, C)) ** float(rng.choice([6, 12]))).astype(np.float32)
    dets, labels, counts = S.batched_multiclass_nms_rotated(torch.from_numpy(bb).cuda(), torch.from_numpy(sc).cuda(), 0.05, 0.5, 2000,
                                                            max_candidates=None if rng.random() < 0.5 else 40000 * B)
    ok = True
    for b in range(B):
        rd, rl = oracle.multiclass_nms_rotated(bb[b], sc[b], 0.05, 0.5, 2000)
        kb = int(counts[b])
        ok &= kb == len(rd) and np.array_equal(dets[b, :kb].cpu().numpy(), rd) and np.array_equal(labels[b, :kb].cpu().numpy().astype(np.float32), rl)
    return bool(ok), "batched B %d n %d" % (B, n)


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    bad, t0 = 0, time.time()
    for c in range(cases):
        ok, msg = nms_case(rng)
        bad += not ok
        print("case %2d %s %s" % (c, msg, "ok" if ok else "MISMATCH"), flush=True)
    for c in range(max(2, cases // 10)):
        ok, msg = batched_case(rng)
        bad += not ok
        print("%s: %s" % (msg, "ok" if ok else "MISMATCH"), flush=True)
    print("mismatches:", bad, "in %.0f s" % (time.time() - t0))
    sys.exit(1 if bad else 0)
