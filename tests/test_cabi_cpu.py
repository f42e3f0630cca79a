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
