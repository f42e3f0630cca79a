"""The 16x16x32 lane maps of the convolution / AlignConv kernels (csrc/dcn_ops.hip: pix16, fbase16, abase16) restated on the host:
every lane's pixel is distinct, the B-fragment and A-fragment ds_read_b128 are conflict-free on the 144-byte rows, and both operands
address the same 8-channel group of the 64-channel chunk.  A device read of 16 B per lane is served in four groups of 16 lanes
({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32); a group is conflict-free when its sixteen 16-byte slots differ mod 16."""
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]
ROW = 144            # kRowBytes: 128 B of channels + 16 B pad


def pix16(i):
    return (((i - 4) >> 1) * 4 + (i & 1)) if 4 <= i < 12 else (((i & 3) >> 1) * 4 + 2 + (i & 1) + (8 if i >= 12 else 0))


def b_offset(lane, ks):          # byte offset of the lane's B fragment inside a 16-position tile (tap / tile offsets are uniform)
    i, kg = lane & 15, lane >> 4
    return pix16(i) * ROW + (kg & 1) * 64 + (kg >> 1) * 16 + ks * 32


def a_offset(lane, ot, ks):      # byte offset inside one out-channel group's 8 KB of the packed (32x32x16-ordered) filter
    i, kg = lane & 15, lane >> 4
    return (kg & 1) * 2048 + ((kg >> 1) * 32 + i) * 16 + (ot >> 1) * 4096 + ks * 1024 + (ot & 1) * 256


def test_pixel_map_is_a_permutation():
    assert sorted(pix16(i) for i in range(16)) == list(range(16))


def test_b_fragment_reads_are_conflict_free():
    for ks in range(2):
        for g in GROUPS:
            assert len({(b_offset(l, ks) // 16) % 16 for l in g}) == 16


def test_a_fragment_reads_are_conflict_free():
    for ot in range(4):
        for ks in range(2):
            for g in GROUPS:
                assert len({(a_offset(l, ot, ks) // 16) % 16 for l in g}) == 16


def test_a_and_b_take_the_same_channel_group_and_cover_the_chunk():
    """packed filter: fragment (a, kk) of a 64-channel group is 1 KB at (a*4 + kk) KB, lane slot = half*32 + row -> element
    (out channel 32a + row, channels 16kk + 8half .. +7); the 16x16x32 A operand must hand lane (i, kg) of tile ot the row
    16*ot + i and the channel group the B operand reads for that lane"""
    seen = set()
    for ot in range(4):
        for ks in range(2):
            for lane in range(64):
                off = a_offset(lane, ot, ks)
                frag, slot = off // 1024, (off % 1024) // 16
                a, kk, half, row = frag // 4, frag % 4, slot // 32, slot % 32
                assert 32 * a + row == 16 * ot + (lane & 15)
                cgroup_a = 2 * kk + half
                cgroup_b = (b_offset(lane, ks) % ROW) // 16
                assert cgroup_a == cgroup_b
                seen.add((32 * a + row, cgroup_a))
    assert len(seen) == 64 * 8          # every (out channel, channel group) of the chunk exactly once per tap


# ---- Winograd kernel (csrc/wino_ops.hip): raw patch with 80-byte pixels, lane (t = lane & 15, kg = lane >> 4) reads pixel
# 2 t + j (j = 0..3) of a patch row, 16 B at kg * 16; staged output rows of 144 B with the even / odd column split
W_PITCH, W_PC = 80, 34


def test_wino_raw_patch_reads_are_conflict_free():
    for j in range(4):
        for row in range(6):
            for g in GROUPS:
                slots = {(((row * W_PC + 2 * (l & 15) + j) * W_PITCH + (l >> 4) * 16) // 16) % 16 for l in g}
                assert len(slots) == 16


def test_wino_dma_slots_cover_the_patch():
    """slot v = pixel * 5 + q: q < 4 are the four 16-byte channel groups of the 32-channel chunk, q = 4 the pad"""
    seen = set()
    for v in range(48 * 64):
        p, q = divmod(v, 5)
        if q < 4 and p < 18 * W_PC:
            assert v * 16 == p * W_PITCH + q * 16
            seen.add((p, q))
    assert len(seen) == 18 * W_PC * 4


def test_wino_output_slots_are_a_permutation():
    cols = sorted(2 * (s & 15) + ((s >> 4) & 1) for s in range(32))
    assert cols == list(range(32))


# ---------------------------------------------------------------- the three-bf16-plane arithmetic of the f32 kernels (round 6)
def _bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 -- what v_cvt_pk_bf16_f32 does for finite values"""
    import numpy as np
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def test_three_bf16_planes_are_exact_and_six_products_are_an_f32_product():
    """k_dcn_x3 / k_dcn_bwd_weight_x3 (csrc/dcn_ops.hip, dcn_bwd_ops.hip: split3): x == hi + mid + lo EXACTLY for finite f32 values
    away from the underflow range, and the six plane products kept (a1 b1, a1 b2, a2 b1, a1 b3, a3 b1, a2 b2) differ from the
    exact product by less than 3 * 2^-24 of it -- the three dropped terms (a2 b3, a3 b2, a3 b3) are below 2^-24 each"""
    import numpy as np
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(200000) * np.exp2(rng.integers(-40, 40, 200000))).astype(np.float32)
    y = (rng.standard_normal(200000) * np.exp2(rng.integers(-40, 40, 200000))).astype(np.float32)

    def split(v):
        hi = _bf16_rne(v)
        r1 = (v - hi).astype(np.float32)              # exact: both are multiples of the same ulp and the difference fits
        mid = _bf16_rne(r1)
        lo = _bf16_rne((r1 - mid).astype(np.float32))
        return hi, mid, lo
    xh, xm, xl = split(x)
    yh, ym, yl = split(y)
    assert np.array_equal((xh.astype(np.float64) + xm.astype(np.float64) + xl.astype(np.float64)), x.astype(np.float64))
    assert np.array_equal((yh.astype(np.float64) + ym.astype(np.float64) + yl.astype(np.float64)), y.astype(np.float64))
    f = np.float64
    six = xh.astype(f) * yh + xh.astype(f) * ym + xm.astype(f) * yh + xh.astype(f) * yl + xl.astype(f) * yh + xm.astype(f) * ym
    exact = x.astype(f) * y.astype(f)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() < 3 * 2.0 ** -24, rel.max()
