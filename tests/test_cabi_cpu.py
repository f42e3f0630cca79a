"""CPU: the C-ABI library loads and exports every symbol include/s2anet_hip.h declares;
host-side wrappers refuse CPU tensors loudly (no fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "s2anet_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(s2a_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from s2anet_amd import _lib
    L = _lib.lib()
    names = header_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), n
        assert n in _lib.SYMBOLS, f"{n} declared in the header but not bound in _lib.SYMBOLS"
    assert set(_lib.SYMBOLS) == set(names)
    assert b"gfx950" in L.s2a_version()


def test_ops_refuse_cpu_tensors():
    import s2anet_amd as S
    b = torch.zeros(4, 5)
    with pytest.raises(NotImplementedError):
        S.box_iou_rotated(b, b)
    with pytest.raises(NotImplementedError):
        S.ml_nms_rotated(b, torch.zeros(4), torch.zeros(4), 0.5)
    with pytest.raises(NotImplementedError):
        S.arf_forward(torch.zeros(2, 2, 1, 3, 3), torch.zeros(1, 3, 3, 8, dtype=torch.uint8))
    conv = S.DeformConv(4, 4, 3, padding=1)
    with pytest.raises(NotImplementedError):
        conv(torch.zeros(1, 4, 5, 5), torch.zeros(1, 18, 5, 5))
    with pytest.raises(ValueError):
        S.deform_conv(torch.zeros(4, 5, 5), torch.zeros(1, 18, 5, 5), conv.weight)


def test_module_surface_matches_reference_names():
    import s2anet_amd as S
    conv = S.DeformConv(8, 16, 3, padding=1)
    assert list(conv.state_dict()) == ["weight"] and conv.weight.shape == (16, 8, 3, 3)
    ac = S.AlignConv(8, 16, 3)
    assert list(ac.state_dict()) == ["deform_conv.weight"]
    oc = S.ORConv2d(16, 2, kernel_size=3, padding=1, arf_config=(1, 8))
    assert sorted(oc.state_dict()) == ["bias", "indices", "weight"]
    assert oc.weight.shape == (2, 16, 1, 3, 3) and oc.bias.shape == (16,)
    assert oc.indices.dtype == torch.uint8 and oc.indices.shape == (1, 3, 3, 8)
    from conftest import golden
    g = golden("head_glue.npz")
    assert (oc.indices.numpy() == g["orconv_indices_1_8"]).all()
    oc8 = S.ORConv2d(16, 2, kernel_size=3, padding=1, arf_config=(8, 8))
    assert (oc8.indices.numpy() == g["orconv_indices_8_8"]).all()


def test_header_is_self_contained_c():
    """include/s2anet_hip.h compiles on its own as C and as C++ (declaration order, no missing types)"""
    import subprocess
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "s2anet_hip.h")
    for lang, cc in (("c", "gcc"), ("c++", "g++")):
        r = subprocess.run([cc, "-fsyntax-only", "-x", lang, "-Wall", hdr], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_argument_checks_return_codes_without_touching_the_gpu():
    """bad shapes / NULL tensors / misalignment are refused with S2A_EINVAL and a message (the reference's TORCH_CHECK
    -> RuntimeError convention) before any HIP call: checkable on a box without a GPU"""
    import ctypes
    from s2anet_amd import _lib
    L = _lib.lib()
    z = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)          # never dereferenced: the checks fail first
    odd = ctypes.c_void_p(18)          # misaligned
    def msg():
        return L.s2a_last_error().decode()
    # conv: channel counts, kernel size, stride, alignment, NULL
    assert L.s2a_conv_nhwc_f16(one, one, one, z, one, 1, 48, 8, 8, 64, 3, 1, 1, z) == -1 and "multiple of 64" in msg()
    assert L.s2a_conv_nhwc_f16(one, one, one, z, one, 1, 64, 8, 8, 64, 5, 1, 1, z) == -1 and "kernel size" in msg()
    assert L.s2a_conv_nhwc_f16(one, one, one, z, one, 1, 64, 8, 8, 64, 3, 3, 1, z) == -1 and "stride" in msg()
    assert L.s2a_conv_nhwc_f16(z, one, one, z, one, 1, 64, 8, 8, 64, 3, 1, 1, z) == -1 and "NULL" in msg()
    assert L.s2a_conv_nhwc_f16(odd, one, one, z, one, 1, 64, 8, 8, 64, 3, 1, 1, z) == -1 and "aligned" in msg()
    assert L.s2a_conv_nhwc_f16(one, one, one, z, one, 0, 64, 8, 8, 64, 3, 1, 1, z) == 0           # empty batch: nothing to do
    # fused bottleneck tail: only the 64 -> 64 -> 256 form; chain arguments go together
    assert L.s2a_conv3x3_tail1x1_f16(one, one, one, one, one, z, one, z, z, z, 0, 1, 128, 64, 256, 8, 8, z) == -1
    assert "64 -> 64 -> 256" in msg()
    assert L.s2a_conv3x3_tail1x1_f16(one, one, one, one, one, z, one, one, z, z, 64, 1, 64, 64, 256, 8, 8, z) == -1
    assert "go together" in msg()
    assert L.s2a_conv3x3_tail1x1_f16(one, one, one, one, one, z, one, one, one, one, 96, 1, 64, 64, 256, 8, 8, z) == -1
    assert "64 or 128" in msg()
    # pyramid launches: level table
    pyr = _lib.Pyramid()
    pyr.n_levels = 0
    assert L.s2a_pyramid_pixels(ctypes.byref(pyr), 1) == -1
    pyr.n_levels = 2
    pyr.height[0], pyr.width[0], pyr.stride[0] = 8, 8, 8.0
    pyr.height[1], pyr.width[1], pyr.stride[1] = 2, 2, 16.0
    assert L.s2a_pyramid_pixels(ctypes.byref(pyr), 3) == 3 * (64 + 4)
    rc = L.s2a_align_conv_pyramid_f16(one, ctypes.cast(one, ctypes.POINTER(ctypes.c_float)), one, one, 1, 256, 256, 1,
                                      ctypes.byref(pyr), z)
    assert rc == -1 and "smaller than kernel" in msg()          # deform_conv_cuda.cpp:100-103 (shape_check)
    # IoU workspace
    assert L.s2a_box_iou_rotated_workspace_bytes(10, 10) > 0
