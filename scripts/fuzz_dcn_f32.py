"""randomised float32 deformable convolution / AlignConv forward: the three-bf16-plane kernel (k_dcn_x3, the default) against
the oracle (bound 1e-4, the north_star's) and against the f32 matrix instruction's kernel (S2A_DCN_F32=mfma32; 1e-5 of the
largest output) -- ragged images on both sides of the 64 / 128-position tile choice, one to ten 32-channel chunks, 64 ... 320
out channels, tame to wild offsets, NCHW and channels-last storage, operands scaled over many binades.  A bounded, seeded
slice runs in the test suite (tests/test_gpu_fuzz.py); more cases by hand:   python scripts/fuzz_dcn_f32.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def fwd_case(rng, max_h=40, max_w=48):
    """one random forward call -> (ok, description)"""
    import s2anet_amd as S, oracle
    B = int(rng.choice([1, 2, 3]))
    C = 32 * int(rng.integers(1, 11))
    O = 64 * int(rng.integers(1, 6))
    H, W = int(rng.integers(3, max_h)), int(rng.integers(3, max_w))
    amp = float(rng.choice([0.3, 1.5, 6.0, 40.0]))
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((O, C, 3, 3)) * 0.05).astype(np.float32)
    off = (rng.standard_normal((B, 18, H, W)) * amp).astype(np.float32)
    if rng.random() < 0.5:                                  # channel scales over 2^-10 ... 2^10, undone by the filter
        sc = np.exp2(rng.integers(-10, 11, (1, C, 1, 1))).astype(np.float32)
        x, w = x * sc, w / sc
    cl = bool(rng.integers(0, 2))
    dev = torch.device("cuda", 0)
    xt = torch.from_numpy(x).to(dev)
    if cl:
        xt = xt.contiguous(memory_format=torch.channels_last)
    offt, wt = torch.from_numpy(off).to(dev), torch.from_numpy(w).to(dev)
    ref = oracle.deform_conv_forward(x, off, w)
    os.environ.pop("S2A_DCN_F32", None)
    a = S.deform_conv(xt, offt, wt, 1, 1, 1, 1, 1).float().cpu().numpy()
    os.environ["S2A_DCN_F32"] = "mfma32"
    try:
        b = S.deform_conv(xt, offt, wt, 1, 1, 1, 1, 1).float().cpu().numpy()
    finally:
        os.environ.pop("S2A_DCN_F32", None)
    scale = max(1.0, float(np.abs(ref).max()))
    e_ref, e_ab = float(np.abs(a - ref).max()) / scale, float(np.abs(a - b).max()) / scale
    ok = e_ref < 1e-4 and e_ab < 1e-5 and np.isfinite(a).all()
    return ok, "B %d C %3d O %3d %2dx%2d amp %4.1f %s: vs oracle %.1e, vs f32 instruction %.1e" % (
        B, C, O, H, W, amp, "NHWC" if cl else "NCHW", e_ref, e_ab)


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
    bad = 0
    for i in range(cases):
        ok, desc = fwd_case(rng)
        print("case %2d %s  %s" % (i, "ok " if ok else "BAD", desc), flush=True)
        bad += not ok
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
