"""Drop-in aliases: make the reference's own Python (``models/detector.py``, ``models/head.py``,
``utils/bbox_nms_rotated.py`` ...) load THIS library's ops under the names of its six pybind11
extension modules (setup.py:31-72) — the reference tree then runs unchanged under
PyTorch-ROCm without building any of its CUDA sources.

    import s2anet_amd.compat as compat; compat.install()
    sys.path.insert(0, "/path/to/S2ANet"); from models.head import S2ANetHead   # reference code

Functions the model never calls and that are not built (modulated DCN backward, PS-RoI pooling) exist so that imports
succeed and raise NotImplementedError when called.
"""
import sys
import types

from . import dcn, orn, rotated

MODULES = ("models.dcn.deform_conv_cuda", "models.dcn.deform_pool_cuda", "models.orn.orn_cuda",
           "utils.box_iou_rotated.box_iou_rotated_cuda", "utils.nms_rotated.nms_rotated_cuda",
           "utils.ml_nms_rotated.ml_nms_rotated_cuda")


def _not_impl(name):
    def f(*a, **k):
        raise NotImplementedError(f"{name} is outside the MI355X inference hot path (SURVEY.md 8(f))")
    f.__name__ = name
    return f


def build_modules():
    m = {}
    d = types.ModuleType("models.dcn.deform_conv_cuda")
    d.deform_conv_forward_cuda = dcn.deform_conv_forward_cuda            # deform_conv_cuda.cpp:689-690
    d.deform_conv_backward_input_cuda = dcn.deform_conv_backward_input_cuda            # :691-693
    d.deform_conv_backward_parameters_cuda = dcn.deform_conv_backward_parameters_cuda  # :694-696
    d.modulated_deform_conv_cuda_forward = dcn.modulated_deform_conv_cuda_forward      # :697-699
    d.modulated_deform_conv_cuda_backward = _not_impl("modulated_deform_conv_cuda_backward")   # :700-701
    m[d.__name__] = d
    p = types.ModuleType("models.dcn.deform_pool_cuda")
    for n in ("deform_psroi_pooling_cuda_forward", "deform_psroi_pooling_cuda_backward"):
        setattr(p, n, _not_impl(n))                                       # deform_pool_cuda.cpp:84-90
    m[p.__name__] = p
    o = types.ModuleType("models.orn.orn_cuda")
    o.arf_forward = orn.arf_forward                                       # vision.cpp:7-12
    o.arf_backward = orn.arf_backward                                     # vision.cpp:9
    o.rie_forward = orn.rie_forward                                       # vision.cpp:10
    o.rie_backward = orn.rie_backward                                     # vision.cpp:11
    m[o.__name__] = o
    b = types.ModuleType("utils.box_iou_rotated.box_iou_rotated_cuda")
    b.box_iou_rotated = rotated.box_iou_rotated                           # box_iou_rotated.h:40-42
    m[b.__name__] = b
    n1 = types.ModuleType("utils.nms_rotated.nms_rotated_cuda")
    n1.nms_rotated = rotated.nms_rotated_raw                              # nms_rotated.h:38-41
    m[n1.__name__] = n1
    n2 = types.ModuleType("utils.ml_nms_rotated.ml_nms_rotated_cuda")
    n2.ml_nms_rotated = rotated.ml_nms_rotated                            # nms_rotated.h:41-44
    m[n2.__name__] = n2
    return m


def install(force=False):
    """register the alias modules in sys.modules (idempotent)"""
    mods = build_modules()
    for name, mod in mods.items():
        if force or name not in sys.modules:
            sys.modules[name] = mod
    return mods
