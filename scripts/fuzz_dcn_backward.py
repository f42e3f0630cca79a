"""randomised deform_conv backward calls at the AlignConv geometry (ragged images, batches, 1-4 channel chunks, out channels
16 ... 256, tame to wild offsets, f32 and f16, a non-zero gradInput / gradWeight, a scale) against the oracle
(oracle.deform_conv_backward = deform_conv_cuda.cpp:262-489 restated): the fused entry points one by one, the one-call
form, and the weight gradient twice (it has to be bit-identical).  A bounded, seeded slice runs in the test suite
(tests/test_gpu_fuzz.py); more cases by hand: python scripts/fuzz_dcn_backward.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def bwd_case(rng, max_h=30, max_w=40):
    """one random backward call -> (ok, description)"""
    import oracle
    from s2anet_amd.dcn import deform_conv_backward_input_cuda, deform_conv_backward_parameters_cuda, _fused_backward
    dev = torch.device("cuda:0")
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f16 = bool(rng.integers(0, 2))
    B = int(rng.choice([1, 2, 3]))
    C = 64 * int(rng.integers(1, 4))
    O = int(rng.choice([32, 64, 96, 256]))
    H, W = int(rng.integers(3, max_h)), int(rng.integers(3, max_w))
    amp = float(rng.choice([0.3, 1.0, 2.5, 6.0]))
    scale = float(rng.choice([1.0, 0.5]))
    mk = lambda *s: rng.standard_normal(s).astype(np.float32)
    xn, wn, on, gn = mk(B, C, H, W), mk(O, C, 3, 3) * 0.1, mk(B, 18, H, W) * amp, mk(B, O, H, W) * 0.5
    base_i, base_w = mk(B, C, H, W), mk(O, C, 3, 3)
    if f16:
        xn, wn, on, gn = (a.astype(np.float16).astype(np.float32) for a in (xn, wn, on, gn))
        base_i = base_i.astype(np.float16).astype(np.float32)
    gx, goff, gw = oracle.deform_conv_backward(xn, on, wn, gn)
    td = torch.float16 if f16 else torch.float32
    x, off, w, go = (cu(a).to(td) for a in (xn, on, wn, gn))
    args = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
    gi, go_ = cu(base_i).to(td), torch.full((B, 18, H, W), 3.0, device=dev, dtype=td)
    deform_conv_backward_input_cuda(x, off, go, gi, go_, w, None, *args, B)
    gws = []
    for _ in range(2):
        gwt = cu(base_w).clone()
        deform_conv_backward_parameters_cuda(x, off, go, gwt, None, None, *args, scale, B)
        gws.append(gwt.cpu().numpy())
    a1, b1, c1 = (t.float().cpu().numpy() for t in _fused_backward(x, off, w, go))
    tol_i, tol_w = (6e-3, 6e-3) if f16 else (1e-4, 1e-4)
    rel = lambda got, ref: float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))
    errs = dict(gin=rel(gi.float().cpu().numpy() - base_i, gx), goff=rel(go_.float().cpu().numpy(), goff),
                gw=rel(gws[0] - base_w, gw * scale), one_gin=rel(a1, gx), one_goff=rel(b1, goff), one_gw=rel(c1, gw))
    tol_gin = tol_i * (3 if f16 else 1)        # f16: the entry point rounds gradInput + base once more
    ok = (errs["gin"] < tol_gin and errs["goff"] < tol_i and errs["gw"] < tol_w and errs["one_gin"] < tol_i
          and errs["one_goff"] < tol_i and errs["one_gw"] < tol_w and np.array_equal(gws[0], gws[1]))
    return bool(ok), "%s B %d C %3d O %3d %2dx%2d amp %.1f scale %.1f: %s" % (
        "f16" if f16 else "f32", B, C, O, H, W, amp, scale, " ".join("%s %.1e" % kv for kv in errs.items()))


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    bad = 0
    for c in range(cases):
        ok, msg = bwd_case(rng)
        bad += not ok
        print("case %2d %s  %s" % (c, "ok" if ok else "MISMATCH", msg), flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
