"""randomised pyramid AlignConv launches (ragged level sizes, batches, tile counts on either side of multiples of the CU count):
the launch with its half-tile tail must equal the launch without it bit for bit, and level 0 must equal the per-level entry
point.  A bounded, seeded slice runs in the test suite (tests/test_gpu_fuzz.py); more cases by hand:
python scripts/fuzz_alignconv.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def setup(seed=11):
    from s2anet_amd.alignconv import pack_weight
    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).cuda().half()
    return g, pack_weight(w, torch.float16)


def align_case(rng, g, wp, max_side=140):
    """one random pyramid launch -> (ok, description)"""
    from s2anet_amd import pyramid as P
    from s2anet_amd.alignconv import align_conv_forward
    B = int(rng.choice([1, 2, 3, 5, 8]))
    h0, w0 = int(rng.integers(40, max_side)), int(rng.integers(40, max_side))
    nl = int(rng.choice([1, 3, 5]))
    sizes = [(max(3, -(-h0 >> i)), max(3, -(-w0 >> i))) for i in range(nl)]
    strides = tuple(8 << i for i in range(nl))
    lay = P.PyramidLayout(B, sizes, strides)
    x = torch.relu(torch.randn(lay.pixels, 256, generator=g)).cuda().half()
    anc = []
    for (h, ww), st in zip(sizes, strides):
        ys, xs = torch.meshgrid(torch.arange(h), torch.arange(ww), indexing="ij")
        a = torch.stack([xs * st + 0.5 * (st - 1) + torch.randn(h, ww, generator=g) * st * 0.3,
                         ys * st + 0.5 * (st - 1) + torch.randn(h, ww, generator=g) * st * 0.3,
                         4 * st * torch.exp(torch.randn(h, ww, generator=g) * 0.4), 4 * st * torch.exp(torch.randn(h, ww, generator=g) * 0.4),
                         torch.rand(h, ww, generator=g) * 3.14159 - 0.785], -1).float()
        anc.append(a.unsqueeze(0).expand(B, -1, -1, -1).reshape(-1, 5))
    anc = torch.cat(anc).cuda().contiguous()
    tiles = sum(B * (-(-h // 8)) * (-(-ww // 16)) for h, ww in sizes)
    outs = {}
    saved = os.environ.get("S2A_DCN_HALF_TAIL")
    try:
        for mode in ("0", "1", None):
            if mode is None: os.environ.pop("S2A_DCN_HALF_TAIL", None)
            else: os.environ["S2A_DCN_HALF_TAIL"] = mode
            outs[mode] = P.align_conv(lay, x, anc, wp, 256).clone()
    finally:
        os.environ.pop("S2A_DCN_HALF_TAIL", None)
        if saved is not None: os.environ["S2A_DCN_HALF_TAIL"] = saved
    ok = torch.equal(outs["0"], outs["1"]) and torch.equal(outs["0"], outs[None])
    # level 0 through the per-level entry point (NHWC in, NHWC out)
    H0, W0 = sizes[0]
    x0 = lay.level(x, 0)
    a0 = lay.rows(anc, 0).view(B, H0, W0, 5)
    ref0 = align_conv_forward(x0, a0, wp, strides[0], relu=True, packed=True, out_channels=256)
    r0, p0 = ref0.permute(0, 2, 3, 1).reshape(-1, 256), lay.level(outs["0"], 0).permute(0, 2, 3, 1).reshape(-1, 256)
    tiles0 = B * (-(-H0 // 8)) * (-(-W0 // 16))
    # (below 128 tiles the per-level entry runs another kernel of the family: the same sums in another order)
    ok0 = torch.equal(r0, p0) if tiles0 >= 128 else bool(((r0.float() - p0.float()).abs().max() < 2e-2).item())
    return bool(ok and ok0), "B %d sizes %s tiles %d (mod 256: %d): tail on/off %s, level 0 vs per-level entry %s" % (
        B, sizes, tiles, tiles % 256, "ok" if ok else "MISMATCH", "ok" if ok0 else "MISMATCH")


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    g, wp = setup()
    bad = 0
    for c in range(cases):
        ok, msg = align_case(rng, g, wp)
        bad += not ok
        print("case %2d %s" % (c, msg), flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
