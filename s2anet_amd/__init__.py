"""s2anet_amd — MI355X-native (gfx950) dense-inference hot path of S2ANet.

Host-side mirror of the reference's operator surface over the C-ABI
library ``libs2anet_hip.so`` (include/s2anet_hip.h).  There is NO CPU fallback: every op
raises if the HIP library is missing or a tensor is not on the GPU.
"""
from . import _lib  # noqa: F401
from .rotated import (box_iou_rotated, nms_rotated, ml_nms_rotated, multiclass_nms_rotated,
                      batched_multiclass_nms_rotated)
from .orn import arf_forward, arf_backward, active_rotating_filter, ORConv2d, RotationInvariantPooling
from .dcn import DeformConv, DeformConvFunction, deform_conv, deform_conv_forward_cuda
from .alignconv import AlignConv

__all__ = [
    "box_iou_rotated", "nms_rotated", "ml_nms_rotated", "multiclass_nms_rotated",
    "batched_multiclass_nms_rotated", "arf_forward", "arf_backward", "active_rotating_filter", "ORConv2d",
    "RotationInvariantPooling", "DeformConv", "DeformConvFunction", "deform_conv",
    "deform_conv_forward_cuda", "AlignConv",
]
