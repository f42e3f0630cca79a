"""Deformable convolution host side (models/dcn/deform_conv.py).

    deform_conv_cuda.deform_conv_forward_cuda(input, weight, offset, output, columns, ones,
        kW, kH, dW, dH, padW, padH, dilationW, dilationH, group, deformable_group, im2col_step)
    DeformConvFunction / deform_conv(input, offset, weight, stride, padding, dilation, groups,
        deformable_groups, im2col_step=64)
    DeformConv(in, out, kernel_size, stride=1, padding=0, dilation=1, groups=1,
        deformable_groups=1, bias=False).forward(x, offset)      parameter name: ``weight``
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.utils import _pair

from . import _lib


def _is_nhwc(t):
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)


def deform_conv_forward_cuda(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW,
                             padH, dilationW, dilationH, group, deformable_group, im2col_step,
                             relu=False):
    """Same positional signature as the reference pybind function
    (models/dcn/src/deform_conv_cuda.cpp:152-157; W-before-H order).  ``output`` is
    caller-allocated and filled in place; ``columns`` / ``ones`` / ``im2col_step`` are accepted
    and ignored (the fused kernel has no columns buffer).  Returns 1."""
    _lib.require_cuda(input, weight, offset, output)
    if input.dim() not in (3, 4):
        raise RuntimeError("3D or 4D input tensor expected but got: %d" % input.dim())
    if weight.dim() != 4:
        raise RuntimeError("4D weight tensor (nOutputPlane,nInputPlane,kH,kW) expected, but got: %d" % weight.dim())
    if weight.size(2) != kH or weight.size(3) != kW:
        raise RuntimeError("kernel size should be consistent with weight")
    squeeze = input.dim() == 3
    if squeeze:
        input, offset, output = input.unsqueeze(0), offset.unsqueeze(0), output.unsqueeze(0)
    nhwc = _is_nhwc(input) and _is_nhwc(output) and input.dtype != torch.float64
    x = input if nhwc else input.contiguous()
    w = weight.contiguous()
    if w.dtype != x.dtype:
        w = w.to(x.dtype)
    off = offset.contiguous()
    if x.dtype == torch.float64:
        off = off.double()                      # the float64 instantiation takes float64 offsets (deform_conv.py:45 casts them)
    elif off.dtype not in (torch.float32, x.dtype):
        off = off.float()
    B, C, H, W = x.shape
    O = w.shape[0]
    if w.shape[1] * group != C:
        raise RuntimeError("invalid number of input planes, expected: %d, but got: %d" % (w.shape[1] * group, C))
    Ho = (H + 2 * padH - (dilationH * (kH - 1) + 1)) // dH + 1
    Wo = (W + 2 * padW - (dilationW * (kW - 1) + 1)) // dW + 1
    if off.shape[0] != B:
        raise RuntimeError("invalid batch size of offset")
    if off.shape[1] != deformable_group * 2 * kH * kW:
        raise RuntimeError("invalid number of channels of offset")
    if off.shape[2] != Ho or off.shape[3] != Wo:
        raise RuntimeError("invalid spatial size of offset, expected height: %d width: %d, but got "
                           "height: %d width: %d" % (Ho, Wo, off.shape[2], off.shape[3]))
    if tuple(output.shape) != (B, O, Ho, Wo):
        raise RuntimeError("output has the wrong shape")
    out = output if (nhwc or output.is_contiguous()) else torch.empty_like(output, memory_format=torch.contiguous_format)
    p = _lib.DcnParams(B, C, H, W, O, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
                       deformable_group, _lib.dtype_code(x, f64=True), _lib.dtype_code(off, f64=True),
                       _lib.LAYOUT_NHWC if nhwc else _lib.LAYOUT_NCHW, int(bool(relu)))
    L = _lib.lib()
    with torch.cuda.device(x.device):
        ws = _lib.workspace(L.s2a_deform_conv_workspace_bytes(p), x.device, "dcn")
        _lib.check(L.s2a_deform_conv_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(off), _lib.ptr(out), p,
                                             _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device)))
    if out is not output:
        output.copy_(out)
    return 1


def _bwd_common(input, offset, gradOutput, weight_like, kW, kH, dW, dH, padW, padH, dilationW, dilationH,
                group, deformable_group, im2col_step):
    """shape_check + views shared by the two backward entry points (deform_conv_cuda.cpp:62-150)"""
    _lib.require_cuda(input, offset, gradOutput, weight_like)
    if input.dim() not in (3, 4):
        raise RuntimeError("3D or 4D input tensor expected but got: %d" % input.dim())
    if weight_like.dim() != 4 or weight_like.size(2) != kH or weight_like.size(3) != kW:
        raise RuntimeError("kernel size should be consistent with weight")
    x = (input if input.dim() == 4 else input.unsqueeze(0)).contiguous()
    off = (offset if offset.dim() == 4 else offset.unsqueeze(0)).to(x.dtype).contiguous()
    go = (gradOutput if gradOutput.dim() == 4 else gradOutput.unsqueeze(0)).to(x.dtype).contiguous()
    B, C, H, W = x.shape
    if weight_like.shape[1] * group != C:
        raise RuntimeError("invalid number of input planes, expected: %d, but got: %d" % (weight_like.shape[1] * group, C))
    Ho = (H + 2 * padH - (dilationH * (kH - 1) + 1)) // dH + 1
    Wo = (W + 2 * padW - (dilationW * (kW - 1) + 1)) // dW + 1
    if off.shape[0] != B:
        raise RuntimeError("invalid batch size of offset")
    if off.shape[1] != deformable_group * 2 * kH * kW:
        raise RuntimeError("invalid number of channels of offset")
    if tuple(off.shape[2:]) != (Ho, Wo):
        raise RuntimeError("invalid spatial size of offset, expected height: %d width: %d, but got "
                           "height: %d width: %d" % (Ho, Wo, off.shape[2], off.shape[3]))
    O = weight_like.shape[0]
    if tuple(go.shape) != (B, O, Ho, Wo):
        raise RuntimeError("invalid size of gradOutput")
    assert B % im2col_step == 0, "im2col step must divide batchsize"

    def params(step):
        return _lib.DcnParams(step, C, H, W, O, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
                              deformable_group, _lib.dtype_code(x, f64=True), _lib.dtype_code(x, f64=True), _lib.LAYOUT_NCHW, 0)
    return x, off, go, (B, C, H, W, O, Ho, Wo), params


def modulated_deform_conv_cuda_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w,
                                       stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                                       deformable_group, with_bias):
    """Same positional signature as the reference pybind function (models/dcn/src/deform_conv_cuda.cpp:491-497; note
    H-before-W here, unlike deform_conv_forward_cuda).  ``output`` is caller-allocated and filled in place; ``ones`` /
    ``columns`` are accepted and ignored.  Returns None as the reference (void)."""
    _lib.require_cuda(input, weight, offset, mask, output)
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")               # TORCH_CHECK :498
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    B, C, H, W = input.shape
    O, Ck, kh, kw = weight.shape
    if kh != kernel_h or kw != kernel_w:
        raise RuntimeError("Input shape and kernel shape wont match: (%d x %d vs %d x %d)." % (kernel_h, kernel_w, kh, kw))
    if C != Ck * group:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (C, Ck * group))
    Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) // stride_h + 1
    Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) // stride_w + 1
    if tuple(offset.shape) != (B, deformable_group * 2 * kh * kw, Ho, Wo) or \
            tuple(mask.shape) != (B, deformable_group * kh * kw, Ho, Wo):
        raise RuntimeError("offset / mask shape does not match the output size")
    if output.numel() != B * O * Ho * Wo:
        raise RuntimeError("output has the wrong number of elements")
    x = input
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    off = offset.to(x.dtype).contiguous()
    msk = mask.to(x.dtype).contiguous()
    b = bias.to(x.dtype).contiguous() if (with_bias and bias is not None and bias.numel()) else None
    out = output if (output.is_contiguous() and output.dtype == x.dtype) else torch.empty((B, O, Ho, Wo), dtype=x.dtype, device=x.device)
    p = _lib.DcnParams(B, C, H, W, O, kernel_w, kernel_h, stride_w, stride_h, pad_w, pad_h, dilation_w, dilation_h, group,
                       deformable_group, _lib.dtype_code(x), _lib.dtype_code(off), _lib.LAYOUT_NCHW, 0)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().s2a_modulated_deform_conv_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(off),
                                                                _lib.ptr(msk), _lib.ptr(out), p, _lib.stream_ptr(x.device)))
    if out is not output:
        output.view(B, O, Ho, Wo).copy_(out)


# bytes of `columns` the unfused paths keep alive per chunk: the reference sizes the chunk by im2col_step alone (64 images
# if the batch has them: 1.2 GB of columns at P3, batch 8, f32 -- written, then re-read twice from HBM).  The chunk is cut to
# the largest divisor of im2col_step whose columns stay in the 256 MB last-level cache (MALL) -- in the weight-gradient path, where
# it pays (2.81 -> 2.43 ms at P3 x 8 f32); the chunks are independent
# (sums over positions), so the results only differ in the order of the f32 sums, as they do between im2col_step values in
# the reference itself.  S2A_DCN_COLUMNS_MB overrides (0: the reference's chunking).
_COLUMNS_CACHE_MB = 192


def _cache_step(step, bytes_per_image):
    """images per chunk of the unfused weight gradient: the largest divisor of im2col_step whose `columns` chunk stays
    inside the last-level cache (S2A_DCN_COLUMNS_MB, 0 = the reference's chunking by im2col_step itself).  A known
    deviation from the reference's partial-sum ORDER for the same im2col_step; same sums."""
    try:
        mb = int(os.environ.get("S2A_DCN_COLUMNS_MB", _COLUMNS_CACHE_MB))
    except ValueError:
        mb = _COLUMNS_CACHE_MB
    if mb <= 0:
        return step
    best = 1
    for d in range(1, step + 1):
        if step % d == 0 and d * bytes_per_image <= mb << 20:
            best = d
    return best


def deform_conv_backward_input_cuda(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH,
                                    dW, dH, padW, padH, dilationW, dilationH, group, deformable_group,
                                    im2col_step):
    """models/dcn/src/deform_conv_cuda.cpp:262-374, same positional signature.  gradInput (caller-zeroed,
    deform_conv.py:88) and gradOffset are filled in place.  AlignConv geometry (3x3, stride 1, pad 1, one group, C % 32 == 0,
    O % 16 == 0, O <= 256): one fused kernel per call, f32 or f16, no `columns`.  Any other geometry, per chunk of
    im2col_step images: columns = weight^T x gradOutput (library GEMM, the reference's addmm_ :323), then
    s2a_deformable_col2im_coord -> gradOffset and s2a_deformable_col2im -> gradInput.  Returns 1."""
    x, off, go, (B, C, H, W, O, Ho, Wo), params = _bwd_common(
        input, offset, gradOutput, weight, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
        deformable_group, im2col_step)
    w = weight.to(x.dtype).contiguous()
    L = _lib.lib()
    step, npos = im2col_step, im2col_step * Ho * Wo
    p = params(step)
    goff = torch.empty_like(off)
    align_geom = ((kW, kH, dW, dH, padW, padH, dilationW, dilationH) == (3, 3, 1, 1, 1, 1, 1, 1) and group == 1
                  and deformable_group == 1 and C % 32 == 0 and O % 16 == 0 and O <= 256 and H >= 3 and W >= 3
                  and not os.environ.get("S2A_DCN_BWD_UNFUSED"))
    # f32 + AlignConv geometry: the fused dataflow on the f32 matrix instruction (s2a_deform_conv_backward_input_f32),
    # accumulating straight into the caller's gradInput when it is the f32 NCHW tensor deform_conv.py:88 allocates
    if align_geom and x.dtype == torch.float32:
        direct = (gradInput.dtype == torch.float32 and gradInput.is_contiguous() and gradInput.numel() == B * C * H * W
                  and gradInput.device == x.device)
        gin = gradInput if direct else torch.zeros((B, C, H, W), dtype=torch.float32, device=x.device)
        direct_off = (gradOffset.dtype == torch.float32 and gradOffset.is_contiguous() and gradOffset.numel() == goff.numel()
                      and gradOffset.device == x.device)
        if direct_off:
            goff = gradOffset
        ws = _lib.workspace(L.s2a_deform_conv_backward_input_f32_workspace_bytes(B, C, H, W, O), x.device, "dcn_bwd")
        with torch.cuda.device(x.device):
            _lib.check(L.s2a_deform_conv_backward_input_f32(_lib.ptr(x), _lib.ptr(off), _lib.ptr(go), _lib.ptr(w), _lib.ptr(gin),
                                                            _lib.ptr(goff), B, C, H, W, O, _lib.ptr(ws), ws.numel(),
                                                            _lib.stream_ptr(x.device)))
        if not direct:
            gradInput.view(B, C, H, W).add_(gin.to(gradInput.dtype))
        if not direct_off:
            gradOffset.view_as(goff).copy_(goff)
        return 1
    # f16 + AlignConv geometry: ONE fused kernel for the whole batch -- column gradient on the matrix cores, consumed in
    # LDS, no `columns` tensor (s2a_deform_conv_backward_input_f16); the chunking by im2col_step has nothing to chunk then
    if align_geom and x.dtype == torch.float16:
        nbytes = L.s2a_deform_conv_backward_input_workspace_bytes(B, C, H, W, O)
        ws = _lib.workspace(nbytes, x.device, "dcn_bwd")
        # the library sums in f32 and hands the gradient over rounded to f16 once (s2a_deform_conv_backward_typed without the weight
        # gradient): the same values as an f32 tensor converted here, without the zero-filled f32 copy and its two passes
        gin16 = torch.empty((B, C, H, W), dtype=torch.float16, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.s2a_deform_conv_backward_typed(_lib.dtype_code(x), _lib.ptr(x), _lib.ptr(off), _lib.ptr(go), _lib.ptr(w),
                                                        _lib.ptr(gin16), _lib.ptr(goff), None, 1.0, B, C, H, W, O, _lib.ptr(ws),
                                                        ws.numel(), _lib.stream_ptr(x.device)))
        gradInput.view(B, C, H, W).add_(gin16.to(gradInput.dtype))
        gradOffset.view_as(goff).copy_(goff)
        return 1
    # accumulator of the scatter: float32 (float32 / float16 columns), float64 for the float64 instantiation
    gin32 = torch.zeros((B, C, H, W), dtype=torch.float64 if x.dtype == torch.float64 else torch.float32, device=x.device)
    wg = w.view(group, O // group, -1)                                  # [g, O/g, C/g*kh*kw]
    # (cache-sized chunks as in deform_conv_backward_parameters_cuda measured 5 % SLOWER here, 10.1 -> 10.6 ms at P3 x 8 f32:
    # this path is bound by the scatter's atomics and the coordinate pass, not by where `columns` lives)
    with torch.cuda.device(x.device):
        st = _lib.stream_ptr(x.device)
        for e in range(B // step):
            sl = slice(e * step, (e + 1) * step)
            g_e = go[sl].transpose(0, 1).reshape(group, O // group, npos)   # [g, O/g, step*Ho*Wo]
            cols = torch.bmm(wg.transpose(1, 2), g_e).reshape(C * kH * kW, npos).contiguous()
            _lib.check(L.s2a_deformable_col2im_coord(_lib.ptr(cols), _lib.ptr(x[sl]), _lib.ptr(off[sl]),
                                                     _lib.ptr(goff[sl]), p, st))
            _lib.check(L.s2a_deformable_col2im(_lib.ptr(cols), _lib.ptr(off[sl]), _lib.ptr(gin32[sl]), p, st))
    gradInput.view(B, C, H, W).add_(gin32.to(gradInput.dtype))          # accumulate, as the reference's atomics do
    gradOffset.view_as(goff).copy_(goff)
    return 1


def deform_conv_backward_parameters_cuda(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH,
                                         padW, padH, dilationW, dilationH, group, deformable_group, scale,
                                         im2col_step):
    """models/dcn/src/deform_conv_cuda.cpp:376-489, same positional signature.  AlignConv geometry (3x3, stride 1, pad 1,
    one group, C % 64 == 0, O % 32 == 0, O <= 256): one fused kernel per call, f32 or f16, no `columns`.  Otherwise, per chunk:
    columns = s2a_deformable_im2col(input, offset), gradWeight += scale * gradOutput x columns^T (library
    GEMM, the reference's addmm_ :455-459).  gradWeight is accumulated in place.  Returns 1."""
    x, off, go, (B, C, H, W, O, Ho, Wo), params = _bwd_common(
        input, offset, gradOutput, gradWeight, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
        deformable_group, im2col_step)
    L = _lib.lib()
    step = im2col_step
    # f16 + AlignConv geometry: ONE fused kernel for the whole batch (columns formed and contracted in LDS, positions as the
    # MFMA's K through transposing LDS reads; every workgroup writes its partial block, k_dcn_bwd_weight_reduce sums them in a
    # fixed order -- no atomics, bit-reproducible): s2a_deform_conv_backward_weight_f16
    fused = (x.dtype == torch.float16 and (kW, kH, dW, dH, padW, padH, dilationW, dilationH) == (3, 3, 1, 1, 1, 1, 1, 1)
             and group == 1 and deformable_group == 1 and C % 64 == 0 and C <= 4096 and O % 32 == 0 and O <= 256 and H >= 3 and W >= 3
             and not os.environ.get("S2A_DCN_BWD_UNFUSED"))
    if fused:
        acc = torch.zeros((O, C, 3, 3), dtype=torch.float32, device=x.device)
        ws = _lib.workspace(L.s2a_deform_conv_backward_weight_workspace_bytes(B, C, H, W, O), x.device, "dcn_bwd")
        with torch.cuda.device(x.device):
            _lib.check(L.s2a_deform_conv_backward_weight_f16(_lib.ptr(x), _lib.ptr(off), _lib.ptr(go), _lib.ptr(acc), B, C, H, W, O,
                                                             _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device)))
        gradWeight.add_((float(scale) * acc).view_as(gradWeight).to(gradWeight.dtype))
        return 1
    # f32 + AlignConv geometry: the same dataflow on v_mfma_f32_16x16x4_f32 (s2a_deform_conv_backward_weight_f32; partial blocks +
    # deterministic reduce as well), scaled and accumulated straight into the caller's gradWeight when it is an f32 contiguous tensor
    fused32 = (x.dtype == torch.float32 and (kW, kH, dW, dH, padW, padH, dilationW, dilationH) == (3, 3, 1, 1, 1, 1, 1, 1)
               and group == 1 and deformable_group == 1 and C % 64 == 0 and C <= 4096 and O % 32 == 0 and O <= 256 and H >= 3 and W >= 3
               and not os.environ.get("S2A_DCN_BWD_UNFUSED"))
    if fused32:
        direct = (gradWeight.dtype == torch.float32 and gradWeight.is_contiguous() and gradWeight.numel() == O * C * 9
                  and gradWeight.device == x.device)
        acc = gradWeight if direct else torch.zeros((O, C, 3, 3), dtype=torch.float32, device=x.device)
        ws = _lib.workspace(L.s2a_deform_conv_backward_weight_f32_workspace_bytes(B, C, H, W, O), x.device, "dcn_bwd")
        with torch.cuda.device(x.device):
            _lib.check(L.s2a_deform_conv_backward_weight_f32(_lib.ptr(x), _lib.ptr(off), _lib.ptr(go), _lib.ptr(acc), float(scale),
                                                             B, C, H, W, O, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device)))
        if not direct:
            gradWeight.add_(acc.view_as(gradWeight).to(gradWeight.dtype))
        return 1
    step = _cache_step(step, C * kH * kW * Ho * Wo * x.element_size())
    npos, p = step * Ho * Wo, params(step)
    cols = torch.empty((C * kH * kW, npos), dtype=x.dtype, device=x.device)
    acc_dt = torch.float64 if x.dtype == torch.float64 else torch.float32
    acc = torch.zeros((group, O // group, (C // group) * kH * kW), dtype=acc_dt, device=x.device)
    with torch.cuda.device(x.device):
        st = _lib.stream_ptr(x.device)
        for e in range(B // step):
            sl = slice(e * step, (e + 1) * step)
            _lib.check(L.s2a_deformable_im2col(_lib.ptr(x[sl]), _lib.ptr(off[sl]), _lib.ptr(cols), p, st))
            g_e = go[sl].transpose(0, 1).reshape(group, O // group, npos)
            acc += torch.bmm(g_e, cols.view(group, -1, npos).transpose(1, 2)).to(acc_dt)
    gradWeight.add_((float(scale) * acc).view_as(gradWeight).to(gradWeight.dtype))
    return 1


def _fused_backward_ok(input, offset, weight, grad_output, stride, padding, dilation, groups, deformable_groups):
    """both gradients in one library call (s2a_deform_conv_backward): AlignConv geometry, f16 / f32, C % 64, O % 32, O <= 256"""
    B, C, H, W = input.shape
    O = weight.shape[0]
    return (input.dtype in (torch.float16, torch.float32) and tuple(weight.shape[2:]) == (3, 3)
            and stride == (1, 1) and padding == (1, 1) and dilation == (1, 1) and groups == 1 and deformable_groups == 1
            and C % 64 == 0 and C <= 4096 and O % 32 == 0 and O <= 256 and H >= 3 and W >= 3 and B > 0
            and offset.shape == (B, 18, H, W) and grad_output.shape == (B, O, H, W)
            and not os.environ.get("S2A_DCN_BWD_UNFUSED") and not os.environ.get("S2A_DCN_BWD_SEPARATE"))


def _fused_backward(input, offset, weight, grad_output):
    """-> (grad_input, grad_offset, grad_weight) as DeformConvFunction.backward returns them (deform_conv.py:73-118: zeroed
    gradInput / gradOffset / gradWeight filled by the two backward entry points, scale 1)"""
    _lib.require_cuda(input, offset, weight, grad_output)
    B, C, H, W = input.shape
    O = weight.shape[0]
    x = input.contiguous()
    dt = x.dtype
    off, go, w = offset.to(dt).contiguous(), grad_output.to(dt).contiguous(), weight.to(dt).contiguous()
    gin = torch.empty((B, C, H, W), dtype=dt, device=x.device)           # overwritten, in the input's own type
    goff = torch.empty((B, 18, H, W), dtype=dt, device=x.device)
    gw = torch.zeros((O, C, 3, 3), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    code = _lib.dtype_code(x)
    ws = _lib.workspace(L.s2a_deform_conv_backward_workspace_bytes(code, B, C, H, W, O), x.device, "dcn_bwd")
    with torch.cuda.device(x.device):
        _lib.check(L.s2a_deform_conv_backward_typed(code, _lib.ptr(x), _lib.ptr(off), _lib.ptr(go), _lib.ptr(w), _lib.ptr(gin),
                                                    _lib.ptr(goff), _lib.ptr(gw), 1.0, B, C, H, W, O, _lib.ptr(ws), ws.numel(),
                                                    _lib.stream_ptr(x.device)))
    return gin, goff.to(offset.dtype), gw.to(weight.dtype)


class DeformConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1, im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
        if not input.is_cuda:
            raise NotImplementedError     # deform_conv.py:58-59
        cur_im2col_step = min(im2col_step, input.shape[0])
        assert (input.shape[0] % cur_im2col_step) == 0, "im2col step must divide batchsize"
        weight = weight.type_as(input)                      # deform_conv.py:45-46
        kH, kW = weight.shape[2], weight.shape[3]
        Ho = (input.size(2) + 2 * padding[0] - (dilation[0] * (kH - 1) + 1)) // stride[0] + 1
        Wo = (input.size(3) + 2 * padding[1] - (dilation[1] * (kW - 1) + 1)) // stride[1] + 1
        if Ho <= 0 or Wo <= 0:
            raise ValueError("convolution input is too small (output would be {}x{})".format(Ho, Wo))
        fmt = torch.channels_last if _is_nhwc(input) else torch.contiguous_format
        output = torch.empty((input.size(0), weight.size(0), Ho, Wo), dtype=input.dtype,
                             device=input.device, memory_format=fmt)
        deform_conv_forward_cuda(input, weight, offset, output, None, None, kW, kH, stride[1],
                                 stride[0], padding[1], padding[0], dilation[1], dilation[0], groups,
                                 deformable_groups, cur_im2col_step)
        ctx.save_for_backward(input, offset, weight)
        ctx.conf = (stride, padding, dilation, groups, deformable_groups, im2col_step)
        return output

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        """models/dcn/deform_conv.py:73-118"""
        input, offset, weight = ctx.saved_tensors
        stride, padding, dilation, groups, deformable_groups, im2col_step = ctx.conf
        if not grad_output.is_cuda:
            raise NotImplementedError
        cur = min(im2col_step, input.shape[0])
        assert (input.shape[0] % cur) == 0, "im2col step must divide batchsize"
        grad_input = grad_offset = grad_weight = None
        kH, kW = weight.shape[2], weight.shape[3]
        args = (kW, kH, stride[1], stride[0], padding[1], padding[0], dilation[1], dilation[0], groups,
                deformable_groups)
        want_in, want_w = ctx.needs_input_grad[0] or ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if want_in and want_w and _fused_backward_ok(input, offset, weight, grad_output, stride, padding, dilation, groups,
                                                     deformable_groups):
            return (*_fused_backward(input, offset, weight, grad_output), None, None, None, None, None, None)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input = torch.zeros_like(input, memory_format=torch.contiguous_format)
            grad_offset = torch.zeros_like(offset, memory_format=torch.contiguous_format)
            deform_conv_backward_input_cuda(input, offset, grad_output, grad_input, grad_offset, weight, None,
                                            *args, cur)
        if ctx.needs_input_grad[2]:
            grad_weight = torch.zeros_like(weight)
            deform_conv_backward_parameters_cuda(input, offset, grad_output, grad_weight, None, None, *args, 1, cur)
        return grad_input, grad_offset, grad_weight, None, None, None, None, None, None


deform_conv = DeformConvFunction.apply


class DeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, deformable_groups=1, bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0, "in_channels {} cannot be divisible by groups {}".format(in_channels, groups)
        assert out_channels % groups == 0, "out_channels {} cannot be divisible by groups {}".format(out_channels, groups)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        # inputs smaller than the kernel are zero-padded and the result cropped (deform_conv.py:254-271)
        input_pad = x.size(2) < self.kernel_size[0] or x.size(3) < self.kernel_size[1]
        if input_pad:
            pad_h = max(self.kernel_size[0] - x.size(2), 0)
            pad_w = max(self.kernel_size[1] - x.size(3), 0)
            x = F.pad(x, (0, pad_w, 0, pad_h), "constant", 0).contiguous()
            offset = F.pad(offset, (0, pad_w, 0, pad_h), "constant", 0).contiguous()
        out = deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation,
                          self.groups, self.deformable_groups)
        if input_pad:
            out = out[:, :, :out.size(2) - pad_h, :out.size(3) - pad_w].contiguous()
        return out
