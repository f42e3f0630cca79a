"""CPU: the alias modules of s2anet_amd.compat carry the reference's pybind names; where the
reference tree is present (build container) its own Python imports cleanly against them."""
import inspect
import os
import sys
import types

import pytest


def test_alias_modules_have_reference_names():
    import s2anet_amd.compat as compat
    mods = compat.build_modules()
    assert set(mods) == set(compat.MODULES)
    d = mods["models.dcn.deform_conv_cuda"]
    sig = inspect.signature(d.deform_conv_forward_cuda)
    assert list(sig.parameters)[:17] == ["input", "weight", "offset", "output", "columns", "ones", "kW", "kH",
                                         "dW", "dH", "padW", "padH", "dilationW", "dilationH", "group",
                                         "deformable_group", "im2col_step"]
    # backward entry points carry the pybind signatures (deform_conv_cuda.cpp:262-268, :376-381)
    assert list(inspect.signature(d.deform_conv_backward_input_cuda).parameters) == [
        "input", "offset", "gradOutput", "gradInput", "gradOffset", "weight", "columns", "kW", "kH", "dW", "dH",
        "padW", "padH", "dilationW", "dilationH", "group", "deformable_group", "im2col_step"]
    assert list(inspect.signature(d.deform_conv_backward_parameters_cuda).parameters) == [
        "input", "offset", "gradOutput", "gradWeight", "columns", "ones", "kW", "kH", "dW", "dH", "padW", "padH",
        "dilationW", "dilationH", "group", "deformable_group", "scale", "im2col_step"]
    # modulated (DCNv2) forward: pybind signature of deform_conv_cuda.cpp:491-497; its backward is not built
    assert list(inspect.signature(d.modulated_deform_conv_cuda_forward).parameters) == [
        "input", "weight", "bias", "ones", "offset", "mask", "output", "columns", "kernel_h", "kernel_w", "stride_h",
        "stride_w", "pad_h", "pad_w", "dilation_h", "dilation_w", "group", "deformable_group", "with_bias"]
    with pytest.raises(NotImplementedError):
        d.modulated_deform_conv_cuda_backward()
    for n in ("rie_forward", "rie_backward"):
        assert callable(getattr(mods["models.orn.orn_cuda"], n))
    assert callable(mods["models.orn.orn_cuda"].arf_forward) and callable(mods["models.orn.orn_cuda"].arf_backward)
    assert callable(mods["utils.box_iou_rotated.box_iou_rotated_cuda"].box_iou_rotated)
    assert callable(mods["utils.nms_rotated.nms_rotated_cuda"].nms_rotated)
    assert callable(mods["utils.ml_nms_rotated.ml_nms_rotated_cuda"].ml_nms_rotated)


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference tree not present")
def test_reference_python_imports_against_aliases():
    import s2anet_amd.compat as compat
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("models", "utils", "cv2", "torchvision")}
    sys.dont_write_bytecode = True
    try:
        for k in list(saved):
            del sys.modules[k]
        compat.install(force=True)
        cv2 = types.ModuleType("cv2")
        cv2.setNumThreads = lambda *a, **k: None
        sys.modules.setdefault("cv2", cv2)
        sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
        sys.path.insert(0, "/root/reference")
        from models.head import S2ANetHead                       # noqa: F401  (reference code)
        from models.dcn import DeformConv as RefDeformConv
        from models.orn import ORConv2d as RefORConv2d
        from utils.bbox_nms_rotated import multiclass_nms_rotated as ref_mc   # noqa: F401
        ref_dc = sys.modules["models.dcn.deform_conv"]   # (models.dcn re-exports a function of that name)
        assert ref_dc.deform_conv_cuda is sys.modules["models.dcn.deform_conv_cuda"]
        head = S2ANetHead(15)
        assert sum(p.numel() for p in head.parameters()) == 4919592
        assert isinstance(head.align_conv.deform_conv, RefDeformConv) and isinstance(head.or_conv, RefORConv2d)
        # the reference head's parameters load into this repo's head unchanged
        from s2anet_amd.head import S2ANetHead as OurHead
        ours = OurHead(15)
        missing, unexpected = ours.load_state_dict(head.state_dict(), strict=True)
        assert not missing and not unexpected
    finally:
        if "/root/reference" in sys.path:
            sys.path.remove("/root/reference")
        for k in [k for k in sys.modules if k.split(".")[0] in ("models", "utils", "cv2", "torchvision")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_reference_checkpoint_loads_by_position(tmp_path):
    """val.py:153-183: official-code checkpoints are matched entry by entry, not by name"""
    import collections
    import torch
    from s2anet_amd.detector import S2ANet, load_reference_checkpoint
    torch.manual_seed(3)
    src = S2ANet(15)
    with torch.no_grad():
        for p in src.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    renamed = collections.OrderedDict((f"module.{i}", v) for i, (k, v) in enumerate(src.state_dict().items()))
    path = tmp_path / "official.pth"
    torch.save({"state_dict": renamed, "meta": {}}, path)
    dst = load_reference_checkpoint(S2ANet(15), str(path))
    for (k1, v1), (k2, v2) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    dst2 = load_reference_checkpoint(S2ANet(15), {"model": src.state_dict()})
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst2.state_dict().values()))
    short = collections.OrderedDict(list(renamed.items())[:-1])
    with pytest.raises(AssertionError):
        load_reference_checkpoint(S2ANet(15), {"state_dict": short})
    with pytest.raises(KeyError):
        load_reference_checkpoint(S2ANet(15), {"weights": {}})


def test_scale_coords_rotated_matches_reference_function():
    """utils/general.py:629-648 run in the build container (tests/golden/formats.npz): letterbox gain / padding with
    and without an explicit ratio_pad, in place, width / height scaled, angle and score untouched"""
    import numpy as np
    import torch
    from conftest import golden
    from s2anet_amd.formats import scale_coords_rotated
    g = golden("formats.npz")
    for tag in "abcd":
        a = g["args_" + tag]
        rp = None if len(a) == 4 else ((a[4], a[5]), (a[6], a[7]))
        t = torch.from_numpy(g["dets"].copy())
        out = scale_coords_rotated((int(a[0]), int(a[1])), t, (int(a[2]), int(a[3])), rp)
        assert out is t                                                      # in place, as the reference
        assert np.array_equal(out.numpy(), g["out_" + tag]), tag


def test_unfused_backward_column_chunks_fit_the_last_level_cache(monkeypatch):
    """deform_conv_backward_parameters_cuda cuts the reference's im2col_step chunk to the largest divisor whose `columns`
    stay under the cache budget (host logic only; the reference's own chunking with S2A_DCN_COLUMNS_MB=0)"""
    from s2anet_amd.dcn import _cache_step
    per_image = 2304 * 128 * 128 * 4                      # P3, 256 x 3 x 3 rows, f32: 151 MB
    assert _cache_step(8, per_image) == 1
    assert _cache_step(8, per_image // 4) == 4            # 38 MB per image: four of them fit 192 MB
    assert _cache_step(6, per_image // 4) == 3            # divisors only
    assert _cache_step(8, 4 * per_image) == 1             # never below one image
    monkeypatch.setenv("S2A_DCN_COLUMNS_MB", "0")
    assert _cache_step(8, per_image) == 8
