"""soak of the two calls of tests/test_gpu_ops.py::test_nms_random_vs_oracle (ml_nms_rotated and the single-class
nms_rotated_raw on the same rows) with changing sizes and a churned caching allocator; results are compared with the
first call on the same rows (and with the oracle for the first size).  python scripts/soak_nms.py [iterations] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def main():
    import s2anet_amd as S, oracle
    from s2anet_amd.rotated import nms_rotated_raw
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    dev = torch.device("cuda", 0)
    cases = []
    for n, span, nl in ((6000, 500, 15), (4097, 300, 1), (9000, 700, 15), (3000, 120, 3), (12000, 900, 40)):
        d = np.concatenate([rng.uniform(0, span, (n, 2)), rng.uniform(4, 60, (n, 2)), rng.uniform(-0.8, 2.4, (n, 1))], 1).astype(np.float32)
        s = (rng.permutation(n).astype(np.float32) + 1) / (n + 1)
        lab = rng.integers(0, nl, n).astype(np.float32)
        D, Sc, Lb = (torch.from_numpy(a).to(dev) for a in (d, s, lab))
        k0 = S.ml_nms_rotated(D, Sc, Lb, 0.5).cpu()
        k1 = nms_rotated_raw(D, Sc, 0.3).cpu()
        if n == 6000:
            assert np.array_equal(k0.numpy(), oracle.ml_nms_rotated(d, s, lab, 0.5, rule=oracle.RULE_GT, sort_mode=oracle.SORT_GPU, cull=True))
            assert np.array_equal(k1.numpy(), oracle.nms_rotated(d, s, 0.3, rule=oracle.RULE_GT, sort_mode=oracle.SORT_GPU, cull=True))
        cases.append((D, Sc, Lb, k0, k1))
    junk, bad, t0 = [], 0, time.time()
    for i in range(iters):
        D, Sc, Lb, k0, k1 = cases[int(rng.integers(0, len(cases)))]
        junk.append(torch.full((int(rng.integers(1, 1 << 22)),), float("nan"), device=dev))     # churn + poison recycled blocks
        if len(junk) > 6: del junk[int(rng.integers(0, len(junk)))]
        a = S.ml_nms_rotated(D, Sc, Lb, 0.5)
        b = nms_rotated_raw(D, Sc, 0.3)
        if not (torch.equal(a.cpu(), k0) and torch.equal(b.cpu(), k1)):
            bad += 1
            print("MISMATCH at iteration", i, flush=True)
        if i % 100 == 99: print("%d iterations, %d mismatches, %.0f s" % (i + 1, bad, time.time() - t0), flush=True)
    print("soak: %d iterations, %d mismatches" % (iters, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
