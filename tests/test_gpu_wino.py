"""The Winograd F(2,3)-along-x form of the head's 3x3 convolutions (csrc/wino_ops.hip, s2a_conv3x3_wino_pyramid_f16)
against torch's f32 convolution on the same f16 operands, against the direct kernel, per level == pyramid-packed, and the
fused orientation max-pool.  Reference: models/head.py:163-222 (nn.Conv2d 3x3 -> cuDNN; f16 in / f32 accumulate)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(2, 256, 128, 128, 256), (1, 64, 20, 37, 128), (3, 128, 8, 8, 64), (1, 256, 5, 3, 320),
                                   (2, 32, 33, 65, 64), (1, 96, 16, 32, 192)])
def test_wino_conv3x3_f16_vs_torch(shape):
    """the four shapes of test_own_conv3x3_f16_vs_torch (+ a 32-channel layer, a non-power-of-two channel count and sizes
    one past a tile edge): same tolerance against the f32 convolution as the direct kernel; bias, ReLU, no bias"""
    from s2anet_amd.fused import conv_f16, conv_pack_weight, conv_wino_f16, conv_wino_pack_weight
    B, C, H, W, O = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, C, H, W, generator=g).to(dev()).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(O, C, 3, 3, generator=g) * 0.03).to(dev()).half()
    b = torch.randn(O, generator=g).to(dev()).half()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b.float(), padding=1)
    wp = conv_wino_pack_weight(w)
    for relu in (False, True):
        out = conv_wino_f16(x, wp, b, O, relu)
        r = torch.relu(ref) if relu else ref
        assert out.shape == r.shape
        err = (out.float() - r).abs()
        assert err.max().item() < 2e-2 and err.mean().item() < 2e-3, (err.max().item(), err.mean().item())
    out_nb = conv_wino_f16(x, wp, None, O, False)
    assert (out_nb.float() - torch.nn.functional.conv2d(x.float(), w.float(), None, padding=1)).abs().max().item() < 2e-2
    if C % 64 == 0:
        # against the direct kernel: both round the same f32-accumulated sums to f16; the Winograd operands carry one extra
        # f16 rounding each -> a few output ulps
        d = conv_f16(x, conv_pack_weight(w), b, O, 3, 1, False)
        assert (out_nb.float() + b.float().view(1, -1, 1, 1) - d.float()).abs().max().item() < 2e-2
    # deterministic
    assert torch.equal(conv_wino_f16(x, wp, b, O, True), conv_wino_f16(x, wp, b, O, True))


def test_wino_error_is_of_the_size_of_the_direct_kernels():
    """on the head's own shape the Winograd form's error against the f32 convolution stays within 2x the direct kernel's
    (mean) -- the bound the f16 fixture test's unchanged tolerances rest on"""
    from s2anet_amd.fused import conv_f16, conv_pack_weight, conv_wino_f16, conv_wino_pack_weight
    g = torch.Generator().manual_seed(3)
    x = torch.relu(torch.randn(2, 256, 64, 64, generator=g)).to(dev()).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.01).to(dev()).half()
    b = torch.zeros(256).to(dev()).half()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), None, padding=1)
    ew = (conv_wino_f16(x, conv_wino_pack_weight(w), b, 256, False).float() - ref).abs()
    ed = (conv_f16(x, conv_pack_weight(w), b, 256, 3, 1, False).float() - ref).abs()
    assert ew.mean().item() < 2.0 * ed.mean().item() + 1e-6, (ew.mean().item(), ed.mean().item())
    assert ew.max().item() < 3.0 * ed.max().item() + 1e-6, (ew.max().item(), ed.max().item())


def _pyr(B=2, sizes=((40, 56), (20, 28), (10, 14), (5, 7), (3, 4)), C=256):
    from s2anet_amd.pyramid import PyramidLayout
    layout = PyramidLayout(B, sizes, (8, 16, 32, 64, 128)[:len(sizes)])
    g = torch.Generator().manual_seed(5)
    x = torch.randn(layout.pixels, C, generator=g).to(dev()).half()
    return layout, x, g


@pytest.mark.parametrize("sizes", [((40, 56), (20, 28), (10, 14), (5, 7), (3, 4)), ((48, 48), (24, 24), (12, 12), (6, 6), (3, 3)),
                                   ((128, 128), (64, 64), (32, 32), (16, 16), (8, 8))])
def test_wino_pyramid_equals_per_level(sizes):
    """one pyramid-packed launch == the per-level launches of the same kernel (bit-identical), and == torch"""
    from s2anet_amd import pyramid as P
    from s2anet_amd.fused import conv_wino_f16, conv_wino_pack_weight
    layout, x, g = _pyr(sizes=sizes)
    for (C, O) in ((256, 256), (256, 64), (32, 256)):
        xx = x[:, :C].contiguous()
        w = (torch.randn(O, C, 3, 3, generator=g) * 0.03).to(dev()).half()
        b = torch.randn(O, generator=g).to(dev()).half()
        wp = conv_wino_pack_weight(w)
        out = P.conv3x3_wino(layout, xx, wp, b, O, relu=True)
        for l in range(len(layout.sizes)):
            xl = layout.level(xx, l)
            ref = torch.relu(torch.nn.functional.conv2d(xl.float(), w.float(), b.float(), padding=1))
            got = layout.level(out, l)
            assert (got.float() - ref).abs().max().item() < 3e-2
            assert torch.equal(got, conv_wino_f16(xl.contiguous(memory_format=torch.channels_last), wp, b, O, True))


def test_wino_orconv_with_fused_pooling():
    """conv + orientation max-pool in one launch == the conv launch followed by the pooling kernel, bit for bit"""
    from s2anet_amd import pyramid as P
    from s2anet_amd.fused import conv_wino_pack_weight
    layout, x, g = _pyr()
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev()).half()
    b = torch.randn(256, generator=g).to(dev()).half()
    wp = conv_wino_pack_weight(w)
    ref = P.conv3x3_wino(layout, x, wp, b, 256, relu=False)
    out, pooled = P.conv3x3_wino(layout, x, wp, b, 256, relu=False, pool=True)
    assert torch.equal(out, ref) and torch.equal(pooled, P.rot_inv_pool(ref, 8))
    assert torch.equal(pooled, ref.view(-1, 32, 8).max(dim=2)[0])


def test_wino_argument_checks():
    from s2anet_amd import _lib
    from s2anet_amd import pyramid as P
    from s2anet_amd.fused import conv_wino_pack_weight
    L = _lib.lib()
    assert L.s2a_conv_wino_packed_elems(256, 256) == 256 * 256 * 12
    assert L.s2a_conv_wino_packed_elems(100, 256) == -1 and L.s2a_conv_wino_packed_elems(64, 48) == -1
    layout, x, g = _pyr(C=48)
    w = torch.zeros(64, 64, 3, 3, device=dev()).half()
    with pytest.raises(RuntimeError):
        P.conv3x3_wino(layout, x, conv_wino_pack_weight(w), None, 64, relu=False)      # 48 channels: not a multiple of 32
    # batch 0: nothing to do, no launch
    from s2anet_amd.pyramid import PyramidLayout
    lay0 = PyramidLayout(0, [(8, 8)], [8.0])
    P.conv3x3_wino(lay0, torch.empty((0, 64), device=dev()).half(), conv_wino_pack_weight(w), None, 64, relu=False)


def test_head_tower_switch(monkeypatch):
    """S2A_CONV_WINO=1 puts the head's regular 3x3 layers on the Winograd kernel; both forms agree within f16 noise"""
    from s2anet_amd.detector import build_synthetic_detector
    m = build_synthetic_detector(device=dev())
    g = torch.Generator().manual_seed(1)
    imgs = (torch.rand(2, 3, 384, 512, generator=g)).to(dev()).half().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = m.features_to_pred(imgs)
        monkeypatch.setenv("S2A_CONV_WINO", "1")
        got = m.features_to_pred(imgs)
    differs = False
    for gl, rl in zip(got, ref):
        for a, b in zip(gl, rl):
            if a is None:
                continue
            d = (a.float() - b.float()).abs().max().item()
            differs |= d > 0
            assert d < 2e-2 * max(1.0, b.float().abs().max().item()), d
    assert differs            # the switch really selects another kernel
