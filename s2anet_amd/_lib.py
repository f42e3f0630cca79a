"""ctypes binding of libs2anet_hip.so (C ABI declared in include/s2anet_hip.h).

PyTorch is used here only as plumbing: device memory (tensors own the buffers and the
caching allocator provides the workspaces) and the current HIP stream.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("S2A_LIB_PATH") or os.path.join(_HERE, "libs2anet_hip.so")   # override: A/B builds

OK, EINVAL, EWORKSPACE, EHIP, ENOTIMPL = 0, -1, -2, -3, -4
DTYPE_F32, DTYPE_F16, DTYPE_F64 = 0, 1, 2
LAYOUT_NCHW, LAYOUT_NHWC = 0, 1

c_i64, c_int, c_f32, c_sz, c_vp = (ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_size_t,
                                   ctypes.c_void_p)


class DcnParams(ctypes.Structure):
    _fields_ = [("batch", c_i64), ("channels", c_i64), ("height", c_i64), ("width", c_i64),
                ("out_channels", c_i64),
                ("kW", c_int), ("kH", c_int), ("dW", c_int), ("dH", c_int), ("padW", c_int),
                ("padH", c_int), ("dilationW", c_int), ("dilationH", c_int), ("group", c_int),
                ("deformable_group", c_int), ("dtype", c_int), ("offset_dtype", c_int),
                ("layout", c_int), ("relu", c_int)]


class AlignParams(ctypes.Structure):
    _fields_ = [("batch", c_i64), ("channels", c_i64), ("height", c_i64), ("width", c_i64),
                ("out_channels", c_i64), ("stride", c_f32), ("dtype", c_int), ("layout", c_int),
                ("relu", c_int), ("weight_packed", c_int)]


class Pyramid(ctypes.Structure):
    """s2a_pyramid: FPN level table of a pyramid-packed buffer"""
    _fields_ = [("n_levels", ctypes.c_int32), ("height", ctypes.c_int32 * 8), ("width", ctypes.c_int32 * 8),
                ("stride", c_f32 * 8)]


# every symbol include/s2anet_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "s2a_last_error": (ctypes.c_char_p, []),
    "s2a_version": (ctypes.c_char_p, []),
    "s2a_box_iou_rotated_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "s2a_box_iou_rotated": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "s2a_box_iou_rotated_pairs": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "s2a_polyiou_pairs": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "s2a_polyiou_match": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "s2a_assign_labels_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "s2a_assign_labels": (c_int, [c_vp, c_i64, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_int, c_int, c_int,
                                  c_vp, c_vp, c_sz, c_vp]),
    "s2a_nms_poly_workspace_bytes": (c_sz, [c_i64]),
    "s2a_nms_poly": (c_int, [c_vp, c_i64, ctypes.c_double, c_vp, c_vp, ctypes.POINTER(c_i64), c_vp, c_sz, c_vp]),
    "s2a_nms_rotated_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "s2a_ml_nms_rotated": (c_int, [c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp,
                                   ctypes.POINTER(c_i64), c_vp, c_sz, c_vp]),
    "s2a_nms_rotated": (c_int, [c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, ctypes.POINTER(c_i64), c_vp,
                                c_sz, c_vp]),
    "s2a_nms_rotated_f64_workspace_bytes": (c_sz, [c_i64]),
    "s2a_nms_rotated_f64": (c_int, [c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, ctypes.POINTER(c_i64), c_vp, c_sz, c_vp]),
    "s2a_nms_rotated_segmented": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_int32,
                                          ctypes.c_int32, c_f32, c_vp, c_vp, c_vp, ctypes.c_int32,
                                          c_vp, c_sz, c_vp]),
    "s2a_nms_rotated_segmented_dets": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_int32, ctypes.c_int32, c_f32,
                                               ctypes.c_int32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "s2a_multiclass_candidates_workspace_bytes": (c_sz, [c_i64]),
    "s2a_multiclass_candidates": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_i64, c_vp, c_vp, c_vp,
                                          c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "s2a_arf_forward": (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, c_vp,
                                c_vp]),
    "s2a_arf_backward": (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "s2a_modulated_deform_conv_forward": (c_int, [c_vp] * 6 + [ctypes.POINTER(DcnParams), c_vp]),
    "s2a_rie_forward": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    "s2a_rie_backward": (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp]),
    "s2a_rot_inv_pool": (c_int, [c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_vp]),
    "s2a_deform_conv_workspace_bytes": (c_sz, [ctypes.POINTER(DcnParams)]),
    "s2a_deform_conv_forward": (c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(DcnParams), c_vp,
                                        c_sz, c_vp]),
    "s2a_align_offsets": (c_int, [c_vp, c_i64, c_i64, c_i64, c_f32, c_int, c_vp, c_vp]),
    "s2a_align_conv_workspace_bytes": (c_sz, [ctypes.POINTER(AlignParams)]),
    "s2a_align_conv_forward": (c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(AlignParams), c_vp,
                                       c_sz, c_vp]),
    "s2a_dcn_packed_elems": (c_i64, [c_i64, c_i64, c_int]),
    "s2a_nms_small_stats": (c_int, [ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "s2a_dcn_pack_weight": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp]),
    "s2a_bias_act_nhwc": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp]),
    "s2a_bias_act_nhwc_to": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp]),
    "s2a_conv_pack_weight_f16": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp]),
    "s2a_conv_nhwc_f16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int,
                                  c_int, c_vp]),
    "s2a_debug_read_stamps": (c_int, [c_vp, c_i64]),
    "s2a_build_flags": (c_int, []),
    "s2a_deform_conv_backward_input_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64, c_i64]),
    "s2a_deform_conv_backward_weight_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64, c_i64]),
    "s2a_deform_conv_backward_workspace_bytes": (c_sz, [c_int, c_i64, c_i64, c_i64, c_i64, c_i64]),
    "s2a_deform_conv_backward": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i64, c_i64, c_i64, c_i64,
                                         c_i64, c_vp, c_sz, c_vp]),
    "s2a_deform_conv_backward_typed": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i64, c_i64, c_i64, c_i64,
                                               c_i64, c_vp, c_sz, c_vp]),
    "s2a_deform_conv_backward_input_f32_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64, c_i64]),
    "s2a_deform_conv_backward_weight_f32_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64, c_i64]),
    "s2a_deform_conv_backward_weight_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_f32, c_i64, c_i64, c_i64, c_i64, c_i64,
                                                    c_vp, c_sz, c_vp]),
    "s2a_deform_conv_backward_input_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64,
                                                   c_vp, c_sz, c_vp]),
    "s2a_deform_conv_backward_weight_f16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_sz, c_vp]),
    "s2a_deform_conv_backward_input_f16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64,
                                           c_vp, c_sz, c_vp]),
    "s2a_deformable_im2col": (c_int, [c_vp, c_vp, c_vp, ctypes.POINTER(DcnParams), c_vp]),
    "s2a_deformable_col2im": (c_int, [c_vp, c_vp, c_vp, ctypes.POINTER(DcnParams), c_vp]),
    "s2a_deformable_col2im_coord": (c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(DcnParams), c_vp]),
    "s2a_conv3x3_tail1x1_f16": (c_int, [c_vp] * 10 + [c_i64] * 7 + [c_vp]),
    "s2a_conv1x1_add_up2_f16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp]),
    "s2a_pyramid_pixels": (c_i64, [ctypes.POINTER(Pyramid), c_i64]),
    "s2a_conv3x3_pyramid_f16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int,
                                        ctypes.POINTER(Pyramid), c_vp]),
    "s2a_conv3x3_head_pyramid_f16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int,
                                             ctypes.POINTER(Pyramid), c_vp]),
    "s2a_orconv_pool_pyramid_f16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, ctypes.POINTER(Pyramid), c_vp]),
    "s2a_conv_wino_packed_elems": (c_i64, [c_i64, c_i64]),
    "s2a_conv_wino_pack_weight_f16": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "s2a_conv3x3_wino_pyramid_f16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int,
                                             ctypes.POINTER(Pyramid), c_vp]),
    "s2a_align_conv_pyramid_f16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int,
                                           ctypes.POINTER(Pyramid), c_vp]),
    "s2a_fam_refine_anchors_pyramid": (c_int, [c_vp, c_i64, c_i64, ctypes.POINTER(Pyramid), c_f32, c_vp, c_vp]),
    "s2a_stem_packed_elems": (c_i64, []),
    "s2a_stem_pack_weight_f16": (c_int, [c_vp, c_vp, c_vp]),
    "s2a_stem_u8_f16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_vp]),
    "s2a_pyramid_candidates_count": (c_i64, [ctypes.POINTER(Pyramid), c_i64]),
    "s2a_pyramid_candidates": (c_int, [c_vp, c_vp, c_vp, c_i64, ctypes.POINTER(Pyramid), c_int, c_i64, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "s2a_rbox_to_poly": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "s2a_delta2bbox_rotated": (c_int, [c_vp, c_vp, c_i64, c_f32, c_vp, c_vp]),
    "s2a_fam_refine_anchors": (c_int, [c_vp, c_i64, c_i64, c_i64, c_f32, c_f32, c_int, c_int,
                                       c_vp, c_vp]),
}


def build(force=False):
    """compile libs2anet_hip.so in-tree (hipcc --offload-arch=gfx950)"""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "-s", "clean"])
    subprocess.check_call(["make", "-C", csrc, "-s", "-j4"])
    return LIB_PATH


_lib = None


def lib():
    """the loaded library; raises (never falls back) when it is missing"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` or `make -C s2anet_amd/csrc` — s2anet_amd has no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        flags = L.s2a_build_flags()
        if flags and not os.environ.get("S2A_ALLOW_MEASURE_BUILD"):
            raise RuntimeError(
                f"{LIB_PATH} holds an object compiled with a measurement / ablation switch (s2a_build_flags() = {flags:#x}): "
                "rebuild with `make -C s2anet_amd/csrc` (scripts that build such objects set S2A_ALLOW_MEASURE_BUILD=1)")
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        msg = lib().s2a_last_error().decode("utf-8", "replace")
        if rc == ENOTIMPL:
            raise NotImplementedError(msg)
        raise RuntimeError(f"s2anet_hip error {rc}: {msg}")


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            # reference: DeformConvFunction raises NotImplementedError on CPU tensors
            # (models/dcn/deform_conv.py:58-59); the product has no CPU path at all
            raise NotImplementedError("s2anet_amd ops run on the GPU only (got a CPU tensor)")


def dtype_code(t, f64=False):
    """f64=True: entry points with a float64 instantiation (deformable convolution generic path, ARF)"""
    if t.dtype == torch.float32:
        return DTYPE_F32
    if t.dtype == torch.float16:
        return DTYPE_F16
    if f64 and t.dtype == torch.float64:
        return DTYPE_F64
    raise TypeError(f"unsupported dtype {t.dtype} (float32 / float16{' / float64' if f64 else ''} only)")


_ws_cache = {}


def workspace(nbytes, device, tag="ws"):
    """grow-only per-(device, stream, tag) scratch buffer from the torch caching allocator"""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
