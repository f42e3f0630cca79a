// Training-side kernels of the deformable convolution: the three device
// functions the reference's backward is built from --
//   deformable_im2col        models/dcn/src/deform_conv_cuda_kernel.cu:189-276   (sampled columns)
//   deformable_col2im        :278-370   (gradient w.r.t. the input: bilinear scatter, atomics)
//   deformable_col2im_coord  :372-464   (gradient w.r.t. the offsets)
// with get_gradient_weight / get_coordinate_weight of :116-187.  The host side
// (s2anet_amd/dcn.py, mirroring deform_conv_cuda.cpp:262-489) chunks the batch by im2col_step and runs
// the two plain GEMMs around them on the library, exactly as the reference does with addmm_.
// Layouts are the reference's: im [S,C,H,W], offset [S, dg*2*kh*kw, Ho, Wo], columns
// [C*kh*kw, S*Ho*Wo] (row = (c*kh + i)*kw + j, column = (s*Ho + h)*Wo + w).  The input gradient always
// accumulates in f32 (the reference adds in the storage type, half atomics included).
// What is in this file:
//   k_def_im2col / k_def_col2im(_tiled) / k_def_col2im_coord   the three device functions, element per thread, any geometry,
//                                                               f16 / f32 / f64 (the path beside the library GEMMs)
//   AlignConv geometry (3x3, stride 1, pad 1, one group), fused, no `columns` tensor:
//   k_dcn_bwd_input (f16)        column gradient on the matrix cores, consumed in LDS; offset gradient; input gradient as a
//                                gather, summed into an [S,H,W,C] f32 accumulator in 128-byte atomic rows
//   k_dcn_bwd_input_f32          the same on the f32 matrix instruction
//   k_dcn_bwd_weight (f16)       columns formed in LDS, contracted over the positions (transposing LDS reads), split-K + reduce
//   k_dcn_bwd_weight_x3          f32 tensors as three bf16 planes per operand on the 16-bit matrix instruction (the default)
//   k_dcn_bwd_weight_f32         the same on the f32 matrix instruction (S2A_BWD_F32_WEIGHT=mfma32)
//   fused_bwd_run + the extern "C" entry points at the end of the file
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace s2a {
namespace {

struct BwdGeom {
  int C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg, S, Ho, Wo;
};

// arithmetic type of the three unfused kernels: the reference instantiates them per scalar type
// (AT_DISPATCH_FLOATING_TYPES_AND_HALF, deform_conv_cuda_kernel.cu:258,352,450) -- double stays double (gradcheck runs on
// it); float and half compute in float
template <typename T>
struct BwdAcc { using type = float; };
template <>
struct BwdAcc<double> { using type = double; };

template <typename T, typename A = float>
__device__ __forceinline__ A bilinear_at(const T* __restrict__ im, int W, int H, A h, A w) {
  // deformable_im2col_bilinear (:83-114); caller has checked h > -1 && w > -1 && h < H && w < W
  int h_low = (int)floor(h), w_low = (int)floor(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  A lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
  A v1 = (h_low >= 0 && w_low >= 0) ? (A)im[h_low * W + w_low] : (A)0;
  A v2 = (h_low >= 0 && w_high <= W - 1) ? (A)im[h_low * W + w_high] : (A)0;
  A v3 = (h_high <= H - 1 && w_low >= 0) ? (A)im[h_high * W + w_low] : (A)0;
  A v4 = (h_high <= H - 1 && w_high <= W - 1) ? (A)im[h_high * W + w_high] : (A)0;
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

// one thread per (c, s, h_col, w_col): writes the kh*kw column entries of that channel/position
template <typename T>
__global__ void k_def_im2col(int64_t n, const T* __restrict__ im, const T* __restrict__ offset, BwdGeom g,
                             T* __restrict__ col) {
  using A = typename BwdAcc<T>::type;
  for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < n;
       index += (int64_t)gridDim.x * blockDim.x) {
    const int w_col = (int)(index % g.Wo);
    const int h_col = (int)((index / g.Wo) % g.Ho);
    const int s = (int)((index / g.Wo / g.Ho) % g.S);
    const int c = (int)(index / g.Wo / g.Ho / g.S);
    const int dgi = c / (g.C / g.dg);
    const int h_in = h_col * g.stride_h - g.pad_h, w_in = w_col * g.stride_w - g.pad_w;
    const int64_t HoWo = (int64_t)g.Ho * g.Wo;
    const T* imp = im + ((int64_t)s * g.C + c) * g.H * g.W;
    const T* offp = offset + ((int64_t)s * g.dg + dgi) * 2 * g.kh * g.kw * HoWo + (int64_t)h_col * g.Wo + w_col;
    T* colp = col + ((int64_t)c * g.kh * g.kw) * g.S * HoWo + ((int64_t)s * g.Ho + h_col) * g.Wo + w_col;
    for (int i = 0; i < g.kh; i++)
      for (int j = 0; j < g.kw; j++) {
        const int t = i * g.kw + j;
        const A oh = (A)offp[(int64_t)(2 * t) * HoWo], ow = (A)offp[(int64_t)(2 * t + 1) * HoWo];
        const A h_im = h_in + i * g.dil_h + oh, w_im = w_in + j * g.dil_w + ow;
        A val = 0;
        if (h_im > -1 && w_im > -1 && h_im < g.H && w_im < g.W) val = bilinear_at<T, A>(imp, g.W, g.H, h_im, w_im);
        colp[(int64_t)t * g.S * HoWo] = (T)val;
      }
  }
}

template <typename A = float>
__device__ __forceinline__ A gradient_weight(A ah, A aw, int h, int w, int H, int W) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return (A)0;
  int hl = (int)floor(ah), wl = (int)floor(aw), hh = hl + 1, wh = wl + 1;
  A weight = 0;
  if (h == hl && w == wl) weight = (h + 1 - ah) * (w + 1 - aw);
  if (h == hl && w == wh) weight = (h + 1 - ah) * (aw + 1 - w);
  if (h == hh && w == wl) weight = (ah + 1 - h) * (w + 1 - aw);
  if (h == hh && w == wh) weight = (ah + 1 - h) * (aw + 1 - w);
  return weight;
}

// one thread per column entry (c, i, j, s, h_out, w_out): scatter into grad_im (f32, atomics)
template <typename T>
__global__ void k_def_col2im(int64_t n, const T* __restrict__ col, const T* __restrict__ offset, BwdGeom g,
                             typename BwdAcc<T>::type* __restrict__ grad_im) {
  using A = typename BwdAcc<T>::type;
  for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < n;
       index += (int64_t)gridDim.x * blockDim.x) {
    const int w_out = (int)(index % g.Wo);
    const int h_out = (int)((index / g.Wo) % g.Ho);
    const int s = (int)((index / g.Wo / g.Ho) % g.S);
    const int j = (int)((index / g.Wo / g.Ho / g.S) % g.kw);
    const int i = (int)((index / g.Wo / g.Ho / g.S / g.kw) % g.kh);
    const int c = (int)(index / g.Wo / g.Ho / g.S / g.kw / g.kh);
    const int dgi = c / (g.C / g.dg);
    const int64_t HoWo = (int64_t)g.Ho * g.Wo;
    const T* offp = offset + ((int64_t)s * g.dg + dgi) * 2 * g.kh * g.kw * HoWo + (int64_t)h_out * g.Wo + w_out;
    const int t = i * g.kw + j;
    const A oh = (A)offp[(int64_t)(2 * t) * HoWo], ow = (A)offp[(int64_t)(2 * t + 1) * HoWo];
    const A ch = h_out * g.stride_h - g.pad_h + i * g.dil_h + oh;
    const A cw = w_out * g.stride_w - g.pad_w + j * g.dil_w + ow;
    const A top = (A)col[index];
    const int cur_h = (int)ch, cur_w = (int)cw;      // truncation toward zero, as the reference (:318-319)
    for (int dy = -2; dy <= 2; dy++)
      for (int dx = -2; dx <= 2; dx++) {
        const int y = cur_h + dy, x = cur_w + dx;
        if (y >= 0 && y < g.H && x >= 0 && x < g.W && fabs(ch - y) < 1 && fabs(cw - x) < 1) {
          const A wgt = gradient_weight<A>(ch, cw, y, x, g.H, g.W);
          atomicAdd(grad_im + (((int64_t)s * g.C + c) * g.H + y) * g.W + x, wgt * top);
        }
      }
  }
}

// The same scatter for the AlignConv geometry (3x3, stride 1, dilation 1): one workgroup = an 8 x 32 tile of
// output positions x 8 channels.  Its contributions land in a (8+2+2*6) x (32+2+2*6) window around the tile:
// they are summed in an LDS copy of that window (ds_add_f32) and the window is flushed once (one global
// atomic per touched cell instead of one per tap and corner: ~9x fewer, row-contiguous); samples that leave
// the window go straight to memory.  The four bilinear corners are addressed directly -- the reference's
// 5 x 5 search (:321-337) visits exactly the cells with |dy| < 1, |dx| < 1, i.e. these four.
constexpr int kC2TH = 8, kC2TW = 32, kC2Halo = 6, kC2Ch = 8;
constexpr int kC2PH = kC2TH + 2 + 2 * kC2Halo, kC2PW = kC2TW + 2 + 2 * kC2Halo;   // 22 x 46

template <typename T>
__global__ __launch_bounds__(256) void k_def_col2im_tiled(const T* __restrict__ col, const T* __restrict__ offset,
                                                          BwdGeom g, float* __restrict__ grad_im) {
  __shared__ float s_patch[kC2Ch][kC2PH * kC2PW];
  const int tid = threadIdx.x;
  const int txn = (g.Wo + kC2TW - 1) / kC2TW, tyn = (g.Ho + kC2TH - 1) / kC2TH;
  int t = blockIdx.x;
  const int tx = t % txn;
  t /= txn;
  const int ty = t % tyn, s = t / tyn;
  const int c0 = blockIdx.y * kC2Ch;
  const int h_out = ty * kC2TH + tid / kC2TW, w_out = tx * kC2TW + tid % kC2TW;
  const int oy = ty * kC2TH - g.pad_h - kC2Halo, ox = tx * kC2TW - g.pad_w - kC2Halo;   // window origin (input coords)
  for (int i = tid; i < kC2Ch * kC2PH * kC2PW; i += 256) (&s_patch[0][0])[i] = 0.f;
  __syncthreads();
  const int64_t HoWo = (int64_t)g.Ho * g.Wo;
  const bool live = h_out < g.Ho && w_out < g.Wo;
  if (live) {
    const int dgi = c0 / (g.C / g.dg);          // the 8 channels share a deformable group (C/dg % 8 == 0, host check)
    const T* offp = offset + ((int64_t)s * g.dg + dgi) * 18 * HoWo + (int64_t)h_out * g.Wo + w_out;
    for (int tap = 0; tap < 9; tap++) {
      const float ch = h_out - g.pad_h + tap / 3 + (float)offp[(int64_t)(2 * tap) * HoWo];
      const float cw = w_out - g.pad_w + tap % 3 + (float)offp[(int64_t)(2 * tap + 1) * HoWo];
      if (ch <= -1 || ch >= g.H || cw <= -1 || cw >= g.W) continue;      // get_gradient_weight: empty
      const int hl = (int)floorf(ch), wl = (int)floorf(cw);
      const float lh = ch - hl, lw = cw - wl;
      const float wgt[4] = {(1 - lh) * (1 - lw), (1 - lh) * lw, lh * (1 - lw), lh * lw};
      for (int cc = 0; cc < kC2Ch; cc++) {
        const int c = c0 + cc;
        if (c >= g.C) break;
        const float top = (float)col[(((int64_t)c * 9 + tap) * g.S + s) * HoWo + (int64_t)h_out * g.Wo + w_out];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int y = hl + (k >> 1), x = wl + (k & 1);
          if (y < 0 || y >= g.H || x < 0 || x >= g.W || wgt[k] == 0.f) continue;
          const int py = y - oy, px = x - ox;
          if (py >= 0 && py < kC2PH && px >= 0 && px < kC2PW)
            atomicAdd(&s_patch[cc][py * kC2PW + px], wgt[k] * top);
          else
            atomicAdd(grad_im + (((int64_t)s * g.C + c) * g.H + y) * g.W + x, wgt[k] * top);
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < kC2Ch * kC2PH * kC2PW; i += 256) {
    const int cc = i / (kC2PH * kC2PW), r = i % (kC2PH * kC2PW);
    const float v = s_patch[cc][r];
    const int y = oy + r / kC2PW, x = ox + r % kC2PW, c = c0 + cc;
    if (v != 0.f && c < g.C && y >= 0 && y < g.H && x >= 0 && x < g.W)
      atomicAdd(grad_im + (((int64_t)s * g.C + c) * g.H + y) * g.W + x, v);
  }
}

template <typename T, typename A = float>
__device__ __forceinline__ A coordinate_weight(A ah, A aw, int H, int W, const T* __restrict__ im, int dir) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return (A)0;
  int hl = (int)floor(ah), wl = (int)floor(aw), hh = hl + 1, wh = wl + 1;
  A weight = 0;
  if (dir == 0) {
    if (hl >= 0 && wl >= 0) weight += -1 * (wl + 1 - aw) * (A)im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) weight += -1 * (aw - wl) * (A)im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) weight += (wl + 1 - aw) * (A)im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) weight += (aw - wl) * (A)im[hh * W + wh];
  } else {
    if (hl >= 0 && wl >= 0) weight += -1 * (hl + 1 - ah) * (A)im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) weight += (hl + 1 - ah) * (A)im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) weight += -1 * (ah - hl) * (A)im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) weight += (ah - hl) * (A)im[hh * W + wh];
  }
  return weight;
}

// one thread per offset element (s, offset channel, h, w): sum over the channels of its deformable group
template <typename T>
__global__ void k_def_col2im_coord(int64_t n, const T* __restrict__ col, const T* __restrict__ im,
                                   const T* __restrict__ offset, BwdGeom g, T* __restrict__ grad_offset) {
  using A = typename BwdAcc<T>::type;
  const int offset_channels = 2 * g.kh * g.kw * g.dg;
  const int cpg = g.C / g.dg;                 // image channels per deformable group
  for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < n;
       index += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(index % g.Wo);
    const int h = (int)((index / g.Wo) % g.Ho);
    const int c = (int)((index / g.Wo / g.Ho) % offset_channels);
    const int s = (int)(index / g.Wo / g.Ho / offset_channels);
    const int dgi = c / (2 * g.kh * g.kw);
    const int offset_c = c - dgi * 2 * g.kh * g.kw;
    const int t = offset_c / 2, dir = offset_c % 2;
    const int i = t / g.kw, j = t % g.kw;
    const int64_t HoWo = (int64_t)g.Ho * g.Wo;
    const T* offp = offset + ((int64_t)s * g.dg + dgi) * 2 * g.kh * g.kw * HoWo + (int64_t)h * g.Wo + w;
    const A oh = (A)offp[(int64_t)(2 * t) * HoWo], ow = (A)offp[(int64_t)(2 * t + 1) * HoWo];
    A inv_h = h * g.stride_h - g.pad_h + i * g.dil_h + oh;
    A inv_w = w * g.stride_w - g.pad_w + j * g.dil_w + ow;
    if (inv_h <= -1 || inv_w <= -1 || inv_h >= g.H || inv_w >= g.W) inv_h = inv_w = -2;
    A val = 0;
    for (int cc = 0; cc < cpg; cc++) {
      const int ch = dgi * cpg + cc;
      const T* imp = im + ((int64_t)s * g.C + ch) * g.H * g.W;
      const A cw = coordinate_weight<T, A>(inv_h, inv_w, g.H, g.W, imp, dir);
      const int64_t col_pos = (((int64_t)ch * g.kh * g.kw + t) * g.S + s) * HoWo + (int64_t)h * g.Wo + w;
      val += cw * (A)col[col_pos];
    }
    grad_offset[index] = (T)val;
  }
}

// the same for f16 with C % 8 == 0, HW % 8 == 0 and 16-byte aligned tensors: 64 x 64 tiles, 16 bytes per lane on both sides
// (eight positions of a channel in, eight channels of a position out) -- the 32 x 32 form moves 2-byte elements in 64-byte rows
// (2.1 TB/s at P3 x 8: 63 us per tensor, two per backward call)
__global__ __launch_bounds__(256) void k_bwd_nchw_to_nhwc_h8(const _Float16* __restrict__ src, int C, int64_t HW,
                                                             _Float16* __restrict__ dst) {
  constexpr int kPitch = 68;                    // halfs per channel row of the tile (136 B: 8-byte aligned vector halves)
  __shared__ __attribute__((aligned(16))) _Float16 tile[64 * kPitch];
  using h4 = __attribute__((ext_vector_type(4))) _Float16;
  using f16x8b = __attribute__((ext_vector_type(8))) _Float16;
  const int64_t b = blockIdx.z, p0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int idx = threadIdx.x + 256 * i, r = idx >> 3, v = idx & 7;
    f16x8b d = {};
    if (c0 + r < C && p0 + v * 8 < HW) d = *reinterpret_cast<const f16x8b*>(src + (b * C + c0 + r) * HW + p0 + v * 8);
    *reinterpret_cast<h4*>(tile + r * kPitch + v * 8) = h4{d[0], d[1], d[2], d[3]};
    *reinterpret_cast<h4*>(tile + r * kPitch + v * 8 + 4) = h4{d[4], d[5], d[6], d[7]};
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int idx = threadIdx.x + 256 * i, pr = idx >> 3, v = idx & 7;
    if (c0 + v * 8 >= C || p0 + pr >= HW) continue;
    f16x8b d;
#pragma unroll
    for (int j = 0; j < 8; j++) d[j] = tile[(v * 8 + j) * kPitch + pr];
    *reinterpret_cast<f16x8b*>(dst + (b * HW + p0 + pr) * C + c0 + v * 8) = d;
  }
}

__global__ __launch_bounds__(256) void k_bwd_zero4(__attribute__((ext_vector_type(4))) float* __restrict__ p, int64_t n4) {
  const __attribute__((ext_vector_type(4))) float z = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) p[i] = z;
}

// the fused input-gradient kernels sum into an [S, HW, C] f32 accumulator (128-byte atomic rows); this hands the result to
// the caller's [S, C, HW] tensor: added to it (ACCUM: the f32 gradInput the reference's atomics accumulate into,
// deform_conv_cuda_kernel.cu:339) or written in the tensor's own type (the zeroed gradInput of deform_conv.py:88)
template <typename OutT, bool ACCUM>
__global__ __launch_bounds__(256) void k_bwd_acc_to_nchw(const float* __restrict__ acc, int C, int64_t HW, OutT* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int64_t b = blockIdx.z, p0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int64_t p = p0 + r;
    const int c = c0 + tx;
    if (c < C && p < HW) tile[r][tx] = acc[(b * HW + p) * C + c];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const int64_t p = p0 + tx;
    if (c < C && p < HW) {
      OutT* d = dst + (b * C + c) * HW + p;
      if constexpr (ACCUM) *d = (OutT)((float)*d + tile[tx][r]);
      else *d = (OutT)tile[tx][r];
    }
  }
}

// ================================================================= fused input + offset gradient (f16, AlignConv geometry)
// deform_conv_backward_input_cuda (models/dcn/src/deform_conv_cuda.cpp:262-374) WITHOUT `columns` in HBM.  The reference
// (and the first version here) materialises columns = W^T x gradOutput [C*9, N] per chunk (604 MB at P3, batch 8), then
// runs col2im_coord and col2im over it: 9-10 ms, all of it memory traffic and global atomics.  Here ONE workgroup owns a
// 4 x 16 tile of output positions and walks the 32-channel chunks of the input:
//   * gradOutput of the tile ([64 pos][O] f16, <= 33 KB) stays in LDS for the whole tile;
//   * per chunk the eight waves form the column-gradient tiles of all nine taps on the MATRIX CORES
//     (v_mfma_f32_32x32x16_f16: G[32 c, 32 pos] = sum_o W[o, c, tap] * gO[o, pos], K = O; filter pre-packed in fragment
//     order, one 1 KB load per k-step) and leave them in LDS as f32 (72 KB);
//   * (position, 8-channel group, tap) items: the four bilinear corners of the sample are read from an LDS patch of the
//     input (the forward's patch idea), four 8-channel dot products give the OFFSET gradient (deformable_col2im_coord's sum
//     over channels, reduced over the four lanes of a position and added to an LDS array);
//   * the INPUT gradient (deformable_col2im's scatter) is a GATHER: the list of (position, tap, corner, weight)
//     contributions of every pixel of a 12 x 24 window is built once per tile, and per chunk a thread sums its pixel's list
//     from the column-gradient tiles -- nine taps and all positions of the tile summed first; the sums of a chunk are parked
//     in LDS and leave as ONE global f32 atomic per touched cell and channel, 128-byte rows of an [S,H,W,C] accumulator
//     (k_bwd_acc_to_nchw hands it to the caller's [S,C,H,W] tensor).
// Samples whose corners leave the window / patch (offsets beyond its 2-3 pixel slack) take global loads and atomics.
// HBM traffic per tile: gradOutput once, the input patch once per chunk, the gradient window once per chunk.
// S2A_BWD_ABL: timing-only ablations of k_dcn_bwd_input (never set in a shipped build): 1 = no global atomics of the gathered
// input gradient, 2 = no gather pass at all, 4 = no offset-gradient pass, 8 = no MFMA jobs, 16 = no list building
// (bits 1 / 2 / 4 / 8 act on k_dcn_bwd_input_f32 too; k_dcn_bwd_weight_f32: 32 = no MFMAs, 64 = no blend, 128 = no tile loads)
#ifdef S2A_MEASURE
// measurement builds: phase cycles of the weight-gradient kernels, summed over workgroups (s2a_debug_bwd_stamps reads and clears).
// k_dcn_bwd_weight_f32: [0] MFMA halves, [1] B1 wait, [2] B2 wait (wave 0); [11..13] the same of wave 4; [3] land + table, [8] wait
// for the loads, [9] land, [7] blend + requests, [10] requests (wave 8); [4] tiles, [5] workgroups, [6] kernel.
// k_dcn_bwd_weight (f16, S2A_MEASURE_F16W): [0] loop-top barrier, [1] land + table, [2] requests, [3] blend, [7] MFMA; [4] [5] [6] as above
__device__ unsigned long long g_bwd_dbg[16];
#define BWD_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define BWD_T(v)
#endif
#ifndef S2A_BWD_ABL
#define S2A_BWD_ABL 0
#endif
constexpr int kBTH = 4, kBTW = 16, kBPos = kBTH * kBTW, kBHalo = 4;
constexpr int kBPH = kBTH + 2 * kBHalo, kBPW = kBTW + 2 * kBHalo, kBPix = kBPH * kBPW;   // 12 x 24 = 288
// k_dcn_bwd_input's window.  (One row and one column less -- 11 x 23 = 253 pixels, so that the 1 012 (8-channel group, pixel)
// items of the gather pass fit ONE trip of the 1 024 threads -- measured 816 -> 884 us at P3 x 8: with 2 instead of 3 pixels of
// slack below / right more AlignConv samples leave the window and take the global path.)
constexpr int kIPH = kBPH, kIPW = kBPW, kIPix = kIPH * kIPW;                             // 12 x 24 = 288
constexpr int kBCh = 32;                       // input channels per chunk (the flush of the window sums takes 32: one 128-byte row per pixel)
constexpr int kBThreads = 1024;                // k_dcn_bwd_input: the passes between the MFMA jobs are latency-bound loops -- sixteen waves
                                               // instead of eight: 1.36 -> 1.34 ms (the f32 kernel: 2.56 -> 2.60, stays at 512)
constexpr int kBGRow = kBCh + 4;               // floats per (tap, position) row of the column-gradient tiles: 144 B -- with 128-B rows every
                                               // lane of the gather pass read the SAME 8 banks (its channel group of a different row):
                                               // the pass took 19 k cycles per chunk for ~2 k cycles of instructions
constexpr int kBWinRow = kBCh + 1;             // window row in floats (+1: the flush reads it pixel-major)
constexpr int kBGoRow = 528;                   // bytes per position of the gradOutput tile: 256 halfs + 16 B pad
using f32x16b = __attribute__((ext_vector_type(16))) float;
using f16x8b = __attribute__((ext_vector_type(8))) _Float16;
using f32x4b = __attribute__((ext_vector_type(4))) float;

struct alignas(16) BTap {
  short y, x;        // top-left bilinear corner (h_low, w_low), image coordinates
  unsigned flags;    // bit 0: sample valid; bit 1: all four corners inside the LDS window; bits 31..2: window pixel index
  _Float16 w[4];     // hh*hw, hh*lw, lh*hw, lh*lw; 0 where the corner is outside the image
};

// weight [O][C][9] f16 -> [tap][C/32][O/16][lane 64][8 halfs]: the A operand of G = W^T . gO
// lane l, element j of fragment (tap, cc, ks) = W[o = ks*16 + 8*(l>>5) + j][c = cc*32 + (l&31)][tap]
__global__ void k_pack_weight_bwd(const _Float16* __restrict__ w, int O, int C, _Float16* __restrict__ wp) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)O * C * 9;
  if (e >= total) return;
  const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
  int64_t r = e >> 9;
  const int KS = O / 16, CC = C / kBCh;
  const int ks = (int)(r % KS);
  r /= KS;
  const int cc = (int)(r % CC), t = (int)(r / CC);
  const int o = ks * 16 + 8 * (lane >> 5) + j, c = cc * kBCh + (lane & 31);
  wp[e] = w[((int64_t)o * C + c) * 9 + t];
}

// 32 x 32 tile transpose: [S, C, HW] -> [S, HW, C]
template <typename T>
__global__ __launch_bounds__(256) void k_bwd_nchw_to_nhwc(const T* __restrict__ src, int C, int64_t HW,
                                                          T* __restrict__ dst) {
  __shared__ T tile[32][33];
  const int64_t b = blockIdx.z, p0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const int64_t p = p0 + tx;
    if (c < C && p < HW) tile[r][tx] = src[(b * C + c) * HW + p];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t p = p0 + r;
    const int c = c0 + tx;
    if (c < C && p < HW) dst[(b * HW + p) * C + c] = tile[tx][r];
  }
}

__global__ __launch_bounds__(kBThreads) void k_dcn_bwd_input(const _Float16* __restrict__ x,        // NHWC [S,H,W,C]
                                                         const _Float16* __restrict__ go,       // NHWC [S,H,W,O]
                                                         const _Float16* __restrict__ offset,   // NCHW [S,18,H,W]
                                                         const _Float16* __restrict__ wpk,
                                                         float* __restrict__ grad_in,           // NHWC [S,H,W,C] f32, accumulated
                                                         _Float16* __restrict__ grad_off,       // NCHW [S,18,H,W]
                                                         int S, int C, int H, int W, int O) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_go = smem;                                                       // [64][kBGoRow]
  float* s_G = reinterpret_cast<float*>(s_go + kBPos * kBGoRow);            // [9][64][kBGRow] f32: column gradient of a chunk
  char* s_patch = reinterpret_cast<char*>(s_G + 9 * kBPos * kBGRow);        // [288][32 halfs]
  BTap* s_tab = reinterpret_cast<BTap*>(s_patch + kIPix * kBCh * 2);        // [64 * 9]
  _Float16* s_frac = reinterpret_cast<_Float16*>(s_tab + kBPos * 9);        // [64 * 9][2]: lh, lw
  float* s_goff = reinterpret_cast<float*>(s_frac + kBPos * 9 * 2);         // [64 * 9][2]
  unsigned* s_list = reinterpret_cast<unsigned*>(s_goff + kBPos * 9 * 2);   // [64 * 9 * 4]: (tap * 64 + pos) << 16 | weight (f16 bits)
  unsigned* s_start = s_list + kBPos * 9 * 4;                               // [288 + 1] first list entry of a window pixel
  unsigned* s_cur = s_start + kIPix + 1;                                    // [288] fill cursors
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int txn = (W + kBTW - 1) / kBTW, tyn = (H + kBTH - 1) / kBTH;
  int t_ = blockIdx.x;
  const int tx0 = (t_ % txn) * kBTW;
  t_ /= txn;
  const int ty0 = (t_ % tyn) * kBTH, b = t_ / tyn;
  // window / patch origin: the undeformed samples of the tile and their lower-right corners span rows ty0 - 1 .. ty0 + 5 and
  // columns tx0 - 1 .. tx0 + 17; the 12 x 24 window leaves 2 pixels of slack above / left and 3 below / right of that
  const int oy = ty0 - 3, ox = tx0 - 3;
  const int64_t HW = (int64_t)H * W;
  const int KS = O / 16, CC = C / kBCh;

  // ---- gradOutput tile -> LDS (positions outside the image: zeros), offset-gradient accumulators, sampling table
  for (int v = tid; v < kBPos * (O / 8); v += kBThreads) {
    const int pos = v / (O / 8), ch = v % (O / 8);
    const int y = ty0 + (pos >> 4), xq = tx0 + (pos & 15);
    f16x8b d = {};
    if (y < H && xq < W) d = *reinterpret_cast<const f16x8b*>(go + ((int64_t)b * HW + (int64_t)y * W + xq) * O + ch * 8);
    *reinterpret_cast<f16x8b*>(s_go + pos * kBGoRow + ch * 16) = d;
  }
  for (int e = tid; e < kBPos * 9 * 2; e += kBThreads) s_goff[e] = 0.f;
  for (int e = tid; e <= kIPix; e += kBThreads) s_start[e] = 0u;
  for (int e = tid; e < kBPos * 9; e += kBThreads) {
    const int pos = e / 9, t = e % 9;
    const int y = ty0 + (pos >> 4), xq = tx0 + (pos & 15);
    BTap tp;
    tp.y = 0; tp.x = 0; tp.flags = 0u;
    for (int k = 0; k < 4; k++) tp.w[k] = (_Float16)0.f;
    float lh = 0.f, lw = 0.f;
    if (y < H && xq < W) {
      const _Float16* ob = offset + ((int64_t)b * 18) * HW + (int64_t)y * W + xq;
      const float off_y = (float)ob[(int64_t)(2 * t) * HW], off_x = (float)ob[(int64_t)(2 * t + 1) * HW];
      const float h_im = (float)(y - 1 + t / 3) + off_y, w_im = (float)(xq - 1 + t % 3) + off_x;
      if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {      // (kernel.cu:228 / get_gradient_weight / coordinate_weight)
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        lh = h_im - h_low; lw = w_im - w_low;
        const float hh = 1 - lh, hw = 1 - lw;
        const bool t_ok = h_low >= 0, b_ok = h_low + 1 <= H - 1, l_ok = w_low >= 0, r_ok = w_low + 1 <= W - 1;
        tp.w[0] = (_Float16)((t_ok && l_ok) ? hh * hw : 0.f);
        tp.w[1] = (_Float16)((t_ok && r_ok) ? hh * lw : 0.f);
        tp.w[2] = (_Float16)((b_ok && l_ok) ? lh * hw : 0.f);
        tp.w[3] = (_Float16)((b_ok && r_ok) ? lh * lw : 0.f);
        tp.y = (short)h_low;
        tp.x = (short)w_low;
        const bool in = h_low >= oy && h_low + 1 <= oy + kIPH - 1 && w_low >= ox && w_low + 1 <= ox + kIPW - 1;
        const int py = min(max(h_low - oy, 0), kIPH - 2), px = min(max(w_low - ox, 0), kIPW - 2);
        tp.flags = 1u | (in ? 2u : 0u) | ((unsigned)(py * kIPW + px) << 2);
      }
    }
    s_tab[e] = tp;
    s_frac[2 * e] = (_Float16)lh;
    s_frac[2 * e + 1] = (_Float16)lw;
  }
  __syncthreads();
  // ---- the scatter of deformable_col2im turned into a GATHER: which (position, tap, corner) lands on which window pixel
  // is the same for every channel, so it is sorted out ONCE per tile (2 304 integer LDS atomics) -- every pixel gets the list
  // of its contributions -- and per chunk a thread sums its pixel's list from the column-gradient tiles and issues one
  // global atomic per cell.  (The first version added w * G into an f32 LDS window with ds_add_f32, 32 per item: LDS
  // float atomics retire about a lane every 2-3 cycles, 6.7 ms per P3 x 8 call against 2.5 ms with three quarters of them
  // skipped.)
  for (int e = tid; e < kBPos * 9; e += kBThreads) {
    const BTap tp = s_tab[e];
    if ((tp.flags & 3u) != 3u) continue;
    const int pix = (int)(tp.flags >> 2);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if ((float)tp.w[k] != 0.f) atomicAdd(&s_start[pix + (k >> 1) * kIPW + (k & 1) + 1], 1u);
  }
  __syncthreads();
  if (wave == 0) {                               // inclusive scan of the 288 counts (shifted by one: s_start[p + 1])
    unsigned carry = 0;
    for (int base = 0; base < kIPix; base += 64) {
      const int p = base + lane;
      unsigned v = p < kIPix ? s_start[p + 1] : 0u;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = (unsigned)__shfl_up((int)v, o);
        if (lane >= o) v += u;
      }
      if (p < kIPix) s_start[p + 1] = carry + v;
      carry += (unsigned)__shfl((int)v, 63);
    }
  }
  __syncthreads();
  for (int e = tid; e < kIPix; e += kBThreads) s_cur[e] = s_start[e];
  __syncthreads();
  for (int e = tid; e < kBPos * 9; e += kBThreads) {
    const BTap tp = s_tab[e];
    if ((tp.flags & 3u) != 3u) continue;
    const int pix = (int)(tp.flags >> 2), pos = e / 9, t = e % 9;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if ((float)tp.w[k] != 0.f) {
        const unsigned slot = atomicAdd(&s_cur[pix + (k >> 1) * kIPW + (k & 1)], 1u);
        s_list[slot] = ((unsigned)(t * kBPos + pos) << 16) | (unsigned)__builtin_bit_cast(unsigned short, tp.w[k]);
      }
  }

  // patch of chunk cc: 288 pixels x 4 vectors of 8 channels; vector v -> pixel v >> 2, group v & 3
  constexpr int kPV = (kIPix * 4 + kBThreads - 1) / kBThreads;
  f16x8b pv[kPV];
  auto patch_issue = [&](int cc) {
#pragma unroll
    for (int i = 0; i < kPV; i++) {
      const int v = tid + kBThreads * i, p = v >> 2, q = v & 3;
      pv[i] = f16x8b{};
      if (v < kIPix * 4) {
        const int yy = oy + p / kIPW, xx = ox + p % kIPW;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
          pv[i] = *reinterpret_cast<const f16x8b*>(x + ((int64_t)b * HW + (int64_t)yy * W + xx) * C + cc * kBCh + q * 8);
      }
    }
  };
  auto patch_write = [&]() {
#pragma unroll
    for (int i = 0; i < kPV; i++) {
      const int v = tid + kBThreads * i;
      if (v < kIPix * 4) *reinterpret_cast<f16x8b*>(s_patch + v * 16) = pv[i];
    }
  };
  patch_issue(0);
  for (int cc = 0; cc < CC; cc++) {
    __syncthreads();                             // everybody is done with the previous chunk's patch and column gradient
    patch_write();
    if (cc + 1 < CC) patch_issue(cc + 1);        // (in flight under this chunk's work)
    // ---- column-gradient tiles of the nine taps on the matrix cores: job = tap, both 32-position halves -- nine waves, one trip.
    // (Jobs of one half, 18 over the sixteen waves, read every filter fragment twice: 288 KB per chunk through a CU's
    // 64 B/clk vector-memory path, 7.2 k cycles for 2.5 k of MFMA; one wave with two accumulators reads it once.)
    for (int job = wave; job < ((S2A_BWD_ABL & 8) ? 0 : 9); job += kBThreads / 64) {
      const int t = job;
      const f16x8b* ap = reinterpret_cast<const f16x8b*>(wpk) + ((int64_t)(t * CC + cc) * KS) * 64 + lane;
      const char* bp = s_go + (lane & 31) * kBGoRow + (lane >> 5) * 16;
      f32x16b acc[2];
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[h][r] = 0.f;
      // the filter fragments come from L2: requested in batches of four right in front of their MFMAs they exposed that
      // latency four times per job (11.5 k cycles per chunk for 2.3 k of MFMA) -- a batch is in flight one batch ahead
      constexpr int kAB = 4;                               // (a 1024-thread workgroup has 128 registers per lane)
      f16x8b a0[kAB], a1[kAB];
      auto load_a = [&](int k0, f16x8b (&a)[kAB]) {
#pragma unroll
        for (int k = 0; k < kAB; k++)
          if (k0 + k < KS) a[k] = ap[(int64_t)(k0 + k) * 64];
      };
      auto run = [&](int k0, const f16x8b (&a)[kAB]) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          f16x8b bb[kAB];
#pragma unroll
          for (int k = 0; k < kAB; k++)
            if (k0 + k < KS) bb[k] = *reinterpret_cast<const f16x8b*>(bp + h * 32 * kBGoRow + (k0 + k) * 32);
#pragma unroll
          for (int k = 0; k < kAB; k++)
            if (k0 + k < KS) acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], bb[k], acc[h], 0, 0, 0);
        }
      };
      load_a(0, a0);
      for (int k0 = 0; k0 < KS; k0 += 2 * kAB) {
        if (k0 + kAB < KS) load_a(k0 + kAB, a1);
        run(k0, a0);
        if (k0 + 2 * kAB < KS) load_a(k0 + 2 * kAB, a0);
        if (k0 + kAB < KS) run(k0 + kAB, a1);
      }
      // D rows = channels (4 consecutive per register quad), columns = positions
#pragma unroll
      for (int h = 0; h < 2; h++) {
        float* gp = s_G + (t * kBPos + h * 32 + (lane & 31)) * kBGRow + 4 * (lane >> 5);
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
          const f32x4b v4 = {acc[h][rq * 4], acc[h][rq * 4 + 1], acc[h][rq * 4 + 2], acc[h][rq * 4 + 3]};
          *reinterpret_cast<f32x4b*>(gp + 8 * rq) = v4;
        }
      }
    }
    __syncthreads();
    // ---- offset gradient: items (tap, position, 8-channel group); the four lanes of a position are neighbours
    for (int it = tid; it < ((S2A_BWD_ABL & 4) ? 0 : 9 * kBPos * 4); it += kBThreads) {
      const int t = it >> 8, r = it & 255, pos = r >> 2, q = r & 3;
      const BTap tp = s_tab[pos * 9 + t];
      if (!(tp.flags & 1u)) continue;            // (the four lanes of a position decide alike)
      const float* gp = s_G + (t * kBPos + pos) * kBGRow + q * 8;
      float G[8];
      {
        const f32x4b g0 = *reinterpret_cast<const f32x4b*>(gp), g1 = *reinterpret_cast<const f32x4b*>(gp + 4);
        G[0] = g0[0]; G[1] = g0[1]; G[2] = g0[2]; G[3] = g0[3]; G[4] = g1[0]; G[5] = g1[1]; G[6] = g1[2]; G[7] = g1[3];
      }
      const bool in = (tp.flags & 2u) != 0u;
      const int pix = (int)(tp.flags >> 2);
      f16x8b c4[4];
      if (in) {
        const char* p0 = s_patch + (pix * 4 + q) * 16;
        c4[0] = *reinterpret_cast<const f16x8b*>(p0);
        c4[1] = *reinterpret_cast<const f16x8b*>(p0 + 64);
        c4[2] = *reinterpret_cast<const f16x8b*>(p0 + kIPW * 64);
        c4[3] = *reinterpret_cast<const f16x8b*>(p0 + kIPW * 64 + 64);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int yy = (int)tp.y + (k >> 1), xx = (int)tp.x + (k & 1);
          c4[k] = f16x8b{};
          if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            c4[k] = *reinterpret_cast<const f16x8b*>(x + ((int64_t)b * HW + (int64_t)yy * W + xx) * C + cc * kBCh + q * 8);
        }
      }
      // get_coordinate_weight (kernel.cu:145-187): corners outside the image hold zeros
      float d[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) a = __builtin_fmaf(G[j], (float)c4[k][j], a);
        d[k] = a;
      }
      const float lh = (float)s_frac[2 * (pos * 9 + t)], lw = (float)s_frac[2 * (pos * 9 + t) + 1];
      float gh = (1.f - lw) * (d[2] - d[0]) + lw * (d[3] - d[1]);
      float gw = (1.f - lh) * (d[1] - d[0]) + lh * (d[3] - d[2]);
      gh += __shfl_xor(gh, 1); gw += __shfl_xor(gw, 1);
      gh += __shfl_xor(gh, 2); gw += __shfl_xor(gw, 2);
      if (q == 0) {                              // one owner per (position, tap): plain read-modify-write
        s_goff[2 * (pos * 9 + t)] += gh;
        s_goff[2 * (pos * 9 + t) + 1] += gw;
      }
      if (!in) {                                 // input gradient of a sample that left the window: straight to memory
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float wk = (float)tp.w[k];
          if (wk == 0.f) continue;
          const int yy = (int)tp.y + (k >> 1), xx = (int)tp.x + (k & 1);
          float* gp2 = grad_in + ((int64_t)b * HW + (int64_t)yy * W + xx) * C + cc * kBCh + q * 8;
#pragma unroll
          for (int j = 0; j < 8; j++) atomicAdd(gp2 + j, wk * G[j]);
        }
      }
    }
    // ---- input gradient: every (8-channel group, window pixel) sums its list; consecutive lanes = consecutive pixels of ONE
    // channel group (conflict-free row reads).  The sums do not go to memory from here: a wave's 64 pixels of one channel
    // are 64 different 1 KB rows of the [S,H,W,C] accumulator (and were 2-3 runs of <= 96 B in an [S,C,H,W] one: 452 us of
    // atomics per P3 x 8 call, against 201 us for the same adds at lane-contiguous addresses, a timing-only build of round 6, docs/HISTORY.md).
    // They are parked in LDS -- in the room of the column-gradient tiles, which nobody reads any more -- and leave as
    // 128-byte rows, one pixel's 32 channels per half wave: two full lines per atomic instruction.
    constexpr int kNI = (4 * kIPix + kBThreads - 1) / kBThreads;
    float a8[kNI][8];
#pragma unroll
    for (int it = 0; it < kNI; it++) {
#pragma unroll
      for (int j = 0; j < 8; j++) a8[it][j] = 0.f;
      const int wi = tid + it * kBThreads;
      if (wi >= ((S2A_BWD_ABL & 2) ? 0 : 4 * kIPix)) continue;
      const int q = wi / kIPix, pix = wi % kIPix;
      const unsigned l0 = s_start[pix], l1 = s_start[pix + 1];
      // four list entries per trip: the entry -> row address -> two row reads chain is three dependent LDS round trips, and a
      // pixel in the middle of the tile has 20-40 entries (one entry per trip: 19 k cycles per chunk, all of it latency).
      // Entries past the end repeat the last one with weight 0; the sums run in list order as before.
      for (unsigned l = l0; l < l1; l += 4) {
        unsigned ent[4];
#pragma unroll
        for (int k = 0; k < 4; k++) ent[k] = s_list[min(l + k, l1 - 1)];
        f32x4b g0[4], g1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float* gp = s_G + (ent[k] >> 16) * kBGRow + q * 8;
          g0[k] = *reinterpret_cast<const f32x4b*>(gp);
          g1[k] = *reinterpret_cast<const f32x4b*>(gp + 4);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float wk = l + k < l1 ? (float)__builtin_bit_cast(_Float16, (unsigned short)(ent[k] & 0xffffu)) : 0.f;
#pragma unroll
          for (int j = 0; j < 4; j++) { a8[it][j] = __builtin_fmaf(wk, g0[k][j], a8[it][j]); a8[it][4 + j] = __builtin_fmaf(wk, g1[k][j], a8[it][4 + j]); }
        }
      }
    }
    __syncthreads();                             // every list has been summed: the column-gradient tiles are free
    float* s_sum = s_G;                          // [288 window pixels][kBGRow]: 32 channel sums of the chunk
#pragma unroll
    for (int it = 0; it < kNI; it++) {
      const int wi = tid + it * kBThreads;
      if (wi >= 4 * kIPix) continue;
      float* sp = s_sum + (wi % kIPix) * kBGRow + (wi / kIPix) * 8;
      *reinterpret_cast<f32x4b*>(sp) = f32x4b{a8[it][0], a8[it][1], a8[it][2], a8[it][3]};
      *reinterpret_cast<f32x4b*>(sp + 4) = f32x4b{a8[it][4], a8[it][5], a8[it][6], a8[it][7]};
    }
    __syncthreads();
    for (int i = tid; i < ((S2A_BWD_ABL & 3) ? 0 : kIPix * kBCh); i += kBThreads) {
      const int pix = i >> 5, c = i & 31;
      if (s_start[pix] == s_start[pix + 1]) continue;            // (pixels outside the image have no list: w = 0 there)
      const int yy = oy + pix / kIPW, xx = ox + pix % kIPW;
      atomicAdd(grad_in + ((int64_t)b * HW + (int64_t)yy * W + xx) * C + cc * kBCh + c, s_sum[pix * kBGRow + c]);
    }
  }
  __syncthreads();
  // ---- offset gradient of the tile: [S, 18, H, W], channel 2 t = dy, 2 t + 1 = dx
  for (int i = tid; i < 18 * kBPos; i += kBThreads) {
    const int ch = i / kBPos, pos = i % kBPos;
    const int y = ty0 + (pos >> 4), xq = tx0 + (pos & 15);
    if (y < H && xq < W)
      grad_off[((int64_t)b * 18 + ch) * HW + (int64_t)y * W + xq] = (_Float16)s_goff[2 * (pos * 9 + (ch >> 1)) + (ch & 1)];
  }
}

constexpr int kBwdLds = kBPos * kBGoRow + 9 * kBPos * kBGRow * 4 + kIPix * kBCh * 2 + kBPos * 9 * 16 + kBPos * 9 * 4 +
                        kBPos * 9 * 8 + kBPos * 9 * 4 * 4 + (kIPix + 1) * 4 + kIPix * 4 + 64;   // ~152 KB

// ================================================================= fused input + offset gradient (f32, AlignConv geometry)
// The same dataflow as k_dcn_bwd_input for float32 tensors (the reference trains in f32 unless amp is on: train.py), on the
// f32 matrix instruction v_mfma_f32_16x16x4_f32 (64 FLOP / clk / SIMD: 157 TFLOP/s on the chip, 1/16 of the f16 rate --
// the column gradient of one P3 x 8 call is 155 GFLOP = 1.0 ms at that peak, so here the MFMA jobs are the long phase).
// Operands are twice as wide, so a workgroup owns a 4 x 8 tile (32 positions) with a 12 x 16 window:
//   * gradOutput tile [32 pos][O] f32 in LDS (33 KB), rows padded by 4 floats (a quarter wave's sixteen 16-byte reads of
//     sixteen rows fall into sixteen different 4-bank groups);
//   * jobs = (tap, 16-position half, 16-channel half): 36 per chunk over 8 waves, 9 per SIMD; K = O is walked 16 out
//     channels at a time: lane (n or m = l & 15, kq = l >> 4) holds o = 16 g + 4 kq + j for the j-th MFMA of group g, so that
//     the gradOutput operand is ONE ds_read_b128 and the filter operand ONE 16-byte global load (filter pre-packed in that
//     order) per four MFMAs;
//   * offset-gradient pass and gather pass as in the f16 kernel (items of 8 channels; corner weights, fractions and the
//     list weights in f32).
// grad_in is the caller's gradInput itself (f32, accumulated in place as the reference's atomics do).
constexpr int kFTH = 4, kFTW = 8, kFPos = kFTH * kFTW, kFHalo = 4;
constexpr int kFPH = kFTH + 2 * kFHalo, kFPW = kFTW + 2 * kFHalo, kFPix = kFPH * kFPW;   // 12 x 16 = 192
constexpr int kFCh = 32;                       // input channels per chunk
constexpr int kFGRow = kFCh + 4;               // floats per (tap, position) row of the column-gradient tiles
constexpr int kFPatRow = kFCh + 4;             // floats per window pixel of the input patch

struct FTap {
  short y, x;        // top-left bilinear corner, image coordinates
  unsigned flags;    // bit 0: sample valid; bit 1: all four corners inside the LDS window; bits 31..2: window pixel index
};

// weight [O][C][9] f32 -> [tap][C/32][c half][O/16][lane 64][4]: lane l, element j of group g = W[o = 16 g + 4 (l >> 4) + j][c][tap],
// c = cc * 32 + half * 16 + (l & 15)
__global__ void k_pack_weight_bwd_f32(const float* __restrict__ w, int O, int C, float* __restrict__ wp) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)O * C * 9;
  if (e >= total) return;
  const int j = (int)(e & 3), lane = (int)((e >> 2) & 63);
  int64_t r = e >> 8;
  const int G16 = O / 16, CC = C / kFCh;
  const int g = (int)(r % G16);
  r /= G16;
  const int half = (int)(r & 1);
  r >>= 1;
  const int cc = (int)(r % CC), t = (int)(r / CC);
  const int o = 16 * g + 4 * (lane >> 4) + j, c = cc * kFCh + half * 16 + (lane & 15);
  wp[e] = w[((int64_t)o * C + c) * 9 + t];
}

__host__ __device__ constexpr int bwd_f32_lds_bytes(int O) {
  return kFPos * (O + 4) * 4 + 9 * kFPos * kFGRow * 4 + kFPix * kFPatRow * 4 + kFPos * 9 * 16 + kFPos * 9 * 8 +
         kFPos * 9 * 8 + kFPos * 9 * 8 + kFPos * 9 * 4 * 4 + kFPos * 9 * 4 * 4 + (kFPix + 1) * 4 + kFPix * 4 + 64;
}

template <bool EXACT>                          // EXACT: O % 128 == 0, every batch of eight filter groups is full (no branches in a job)
__global__ __launch_bounds__(512) void k_dcn_bwd_input_f32(const float* __restrict__ x,        // NHWC [S,H,W,C]
                                                          const float* __restrict__ go,       // NHWC [S,H,W,O]
                                                          const float* __restrict__ offset,   // NCHW [S,18,H,W]
                                                          const float* __restrict__ wpk,
                                                          float* __restrict__ grad_in,        // NCHW [S,C,H,W], accumulated
                                                          float* __restrict__ grad_off,       // NCHW [S,18,H,W]
                                                          int S, int C, int H, int W, int O) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int gop = O + 4;                                                    // floats per position of the gradOutput tile
  float* s_go = reinterpret_cast<float*>(smem);                             // [32][gop]
  float* s_G = s_go + kFPos * gop;                                          // [9][32][kFGRow]: column gradient of a chunk
  float* s_patch = s_G + 9 * kFPos * kFGRow;                                // [192][kFPatRow]
  f32x4b* s_w = reinterpret_cast<f32x4b*>(s_patch + kFPix * kFPatRow);      // [32 * 9] corner weights hh*hw, hh*lw, lh*hw, lh*lw
  FTap* s_tab = reinterpret_cast<FTap*>(s_w + kFPos * 9);                   // [32 * 9]
  float* s_frac = reinterpret_cast<float*>(s_tab + kFPos * 9);              // [32 * 9][2]: lh, lw
  float* s_goff = s_frac + kFPos * 9 * 2;                                   // [32 * 9][2]
  unsigned* s_list = reinterpret_cast<unsigned*>(s_goff + kFPos * 9 * 2);   // [32 * 9 * 4]: tap * 32 + pos
  float* s_lw = reinterpret_cast<float*>(s_list + kFPos * 9 * 4);           // [32 * 9 * 4]: weight of the entry
  unsigned* s_start = reinterpret_cast<unsigned*>(s_lw + kFPos * 9 * 4);    // [192 + 1]
  unsigned* s_cur = s_start + kFPix + 1;                                    // [192]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int txn = (W + kFTW - 1) / kFTW, tyn = (H + kFTH - 1) / kFTH;
  int t_ = blockIdx.x;
  const int tx0 = (t_ % txn) * kFTW;
  t_ /= txn;
  const int ty0 = (t_ % tyn) * kFTH, b = t_ / tyn;
  const int oy = ty0 - 3, ox = tx0 - 3;          // 2 pixels of slack above / left of the undeformed samples, 3 below / right
  const int64_t HW = (int64_t)H * W;
  const int G16 = O / 16, CC = C / kFCh;

  // ---- gradOutput tile -> LDS (positions outside the image: zeros), accumulators, sampling table
  for (int v = tid; v < kFPos * (O / 4); v += 512) {
    const int pos = v / (O / 4), ch = v % (O / 4);
    const int y = ty0 + (pos >> 3), xq = tx0 + (pos & 7);
    f32x4b d = {0.f, 0.f, 0.f, 0.f};
    if (y < H && xq < W) d = *reinterpret_cast<const f32x4b*>(go + ((int64_t)b * HW + (int64_t)y * W + xq) * O + ch * 4);
    *reinterpret_cast<f32x4b*>(s_go + pos * gop + ch * 4) = d;
  }
  for (int e = tid; e < kFPos * 9 * 2; e += 512) s_goff[e] = 0.f;
  for (int e = tid; e <= kFPix; e += 512) s_start[e] = 0u;
  for (int e = tid; e < kFPos * 9; e += 512) {
    const int pos = e / 9, t = e % 9;
    const int y = ty0 + (pos >> 3), xq = tx0 + (pos & 7);
    FTap tp;
    tp.y = 0; tp.x = 0; tp.flags = 0u;
    f32x4b w4 = {0.f, 0.f, 0.f, 0.f};
    float lh = 0.f, lw = 0.f;
    if (y < H && xq < W) {
      const float* ob = offset + ((int64_t)b * 18) * HW + (int64_t)y * W + xq;
      const float off_y = ob[(int64_t)(2 * t) * HW], off_x = ob[(int64_t)(2 * t + 1) * HW];
      const float h_im = (float)(y - 1 + t / 3) + off_y, w_im = (float)(xq - 1 + t % 3) + off_x;
      if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {      // (kernel.cu:228 / get_gradient_weight / coordinate_weight)
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        lh = h_im - h_low; lw = w_im - w_low;
        const float hh = 1 - lh, hw = 1 - lw;
        const bool t_ok = h_low >= 0, b_ok = h_low + 1 <= H - 1, l_ok = w_low >= 0, r_ok = w_low + 1 <= W - 1;
        w4[0] = (t_ok && l_ok) ? hh * hw : 0.f;
        w4[1] = (t_ok && r_ok) ? hh * lw : 0.f;
        w4[2] = (b_ok && l_ok) ? lh * hw : 0.f;
        w4[3] = (b_ok && r_ok) ? lh * lw : 0.f;
        tp.y = (short)h_low;
        tp.x = (short)w_low;
        const bool in = h_low >= oy && h_low + 1 <= oy + kFPH - 1 && w_low >= ox && w_low + 1 <= ox + kFPW - 1;
        const int py = min(max(h_low - oy, 0), kFPH - 2), px = min(max(w_low - ox, 0), kFPW - 2);
        tp.flags = 1u | (in ? 2u : 0u) | ((unsigned)(py * kFPW + px) << 2);
      }
    }
    s_tab[e] = tp;
    s_w[e] = w4;
    s_frac[2 * e] = lh;
    s_frac[2 * e + 1] = lw;
  }
  __syncthreads();
  // ---- contribution lists of the window pixels (see k_dcn_bwd_input)
  for (int e = tid; e < kFPos * 9; e += 512) {
    const FTap tp = s_tab[e];
    if ((tp.flags & 3u) != 3u) continue;
    const f32x4b w4 = s_w[e];
    const int pix = (int)(tp.flags >> 2);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (w4[k] != 0.f) atomicAdd(&s_start[pix + (k >> 1) * kFPW + (k & 1) + 1], 1u);
  }
  __syncthreads();
  if (wave == 0) {                               // inclusive scan of the 192 counts (shifted by one: s_start[p + 1])
    unsigned carry = 0;
    for (int base = 0; base < kFPix; base += 64) {
      const int p = base + lane;
      unsigned v = s_start[p + 1];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = (unsigned)__shfl_up((int)v, o);
        if (lane >= o) v += u;
      }
      s_start[p + 1] = carry + v;
      carry += (unsigned)__shfl((int)v, 63);
    }
  }
  __syncthreads();
  for (int e = tid; e < kFPix; e += 512) s_cur[e] = s_start[e];
  __syncthreads();
  for (int e = tid; e < kFPos * 9; e += 512) {
    const FTap tp = s_tab[e];
    if ((tp.flags & 3u) != 3u) continue;
    const f32x4b w4 = s_w[e];
    const int pix = (int)(tp.flags >> 2), pos = e / 9, t = e % 9;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (w4[k] != 0.f) {
        const unsigned slot = atomicAdd(&s_cur[pix + (k >> 1) * kFPW + (k & 1)], 1u);
        s_list[slot] = (unsigned)(t * kFPos + pos);
        s_lw[slot] = w4[k];
      }
  }

  // patch of chunk cc: 192 pixels x 8 vectors of 4 channels; vector v -> pixel v >> 3, group v & 7
  f32x4b pv[3];
  auto patch_issue = [&](int cc) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int v = tid + 512 * i, p = v >> 3, q = v & 7;
      pv[i] = f32x4b{0.f, 0.f, 0.f, 0.f};
      const int yy = oy + p / kFPW, xx = ox + p % kFPW;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W)
        pv[i] = *reinterpret_cast<const f32x4b*>(x + ((int64_t)b * HW + (int64_t)yy * W + xx) * C + cc * kFCh + q * 4);
    }
  };
  auto patch_write = [&]() {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int v = tid + 512 * i;
      *reinterpret_cast<f32x4b*>(s_patch + (v >> 3) * kFPatRow + (v & 7) * 4) = pv[i];
    }
  };
  patch_issue(0);
  for (int cc = 0; cc < CC; cc++) {
    __syncthreads();                             // everybody is done with the previous chunk's patch and column gradient
    patch_write();
    if (cc + 1 < CC) patch_issue(cc + 1);        // (in flight under this chunk's work)
    // ---- column-gradient tiles: job = (tap, 16-position half, 16-channel half); waves 0-3 take five jobs, 4-7 four: nine per SIMD
    for (int job = wave; job < ((S2A_BWD_ABL & 8) ? 0 : 36); job += 8) {
      const int t = job >> 2, ph = (job >> 1) & 1, chh = job & 1;
      const f32x4b* ap = reinterpret_cast<const f32x4b*>(wpk) + ((int64_t)((t * CC + cc) * 2 + chh) * G16) * 64 + lane;
      const float* bp = s_go + (ph * 16 + (lane & 15)) * gop + 4 * (lane >> 4);
      // two accumulators (back-to-back MFMAs on ONE accumulator wait for each other's eight passes), and both operands of a
      // batch of eight groups requested one batch ahead: the filter comes from L2, the gradOutput rows from LDS, and hipcc
      // places a read right in front of its MFMA unless the order is pinned
      f32x4b acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      f32x4b a0[8], a1[8], b0[8], b1[8];
      auto load_ab = [&](int k0, f32x4b (&a)[8], f32x4b (&bb)[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (EXACT || k0 + k < G16) {
            a[k] = ap[(int64_t)(k0 + k) * 64];
            bb[k] = *reinterpret_cast<const f32x4b*>(bp + (k0 + k) * 16);
          }
      };
      auto run = [&](int k0, const f32x4b (&a)[8], const f32x4b (&bb)[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (EXACT || k0 + k < G16) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][0], bb[k][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][1], bb[k][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][2], bb[k][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][3], bb[k][3], acc1, 0, 0, 0);
          }
      };
      load_ab(0, a0, b0);
      for (int k0 = 0; k0 < G16; k0 += 16) {
        if (k0 + 8 < G16) load_ab(k0 + 8, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        run(k0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 16 < G16) load_ab(k0 + 16, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 8 < G16) run(k0 + 8, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      const f32x4b acc = acc0 + acc1;
      // D: register r = channel 4 (lane >> 4) + r of the half, column = position lane & 15
      *reinterpret_cast<f32x4b*>(s_G + (t * kFPos + ph * 16 + (lane & 15)) * kFGRow + chh * 16 + 4 * (lane >> 4)) = acc;
    }
    __syncthreads();
    // ---- offset gradient: items (tap, position, 8-channel group); the four lanes of a position are neighbours
    for (int it = tid; it < ((S2A_BWD_ABL & 4) ? 0 : 9 * kFPos * 4); it += 512) {
      const int t = it >> 7, r = it & 127, pos = r >> 2, q = r & 3;
      const FTap tp = s_tab[pos * 9 + t];
      if (!(tp.flags & 1u)) continue;            // (the four lanes of a position decide alike)
      const float* gp = s_G + (t * kFPos + pos) * kFGRow + q * 8;
      const f32x4b g0 = *reinterpret_cast<const f32x4b*>(gp), g1 = *reinterpret_cast<const f32x4b*>(gp + 4);
      const bool in = (tp.flags & 2u) != 0u;
      const int pix = (int)(tp.flags >> 2);
      f32x4b c0[4], c1[4];
      if (in) {
        const float* p0 = s_patch + pix * kFPatRow + q * 8;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float* pk = p0 + ((k >> 1) * kFPW + (k & 1)) * kFPatRow;
          c0[k] = *reinterpret_cast<const f32x4b*>(pk);
          c1[k] = *reinterpret_cast<const f32x4b*>(pk + 4);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int yy = (int)tp.y + (k >> 1), xx = (int)tp.x + (k & 1);
          c0[k] = f32x4b{0.f, 0.f, 0.f, 0.f};
          c1[k] = c0[k];
          if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const float* pk = x + ((int64_t)b * HW + (int64_t)yy * W + xx) * C + cc * kFCh + q * 8;
            c0[k] = *reinterpret_cast<const f32x4b*>(pk);
            c1[k] = *reinterpret_cast<const f32x4b*>(pk + 4);
          }
        }
      }
      // get_coordinate_weight (kernel.cu:145-187): corners outside the image hold zeros
      float d[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) a = __builtin_fmaf(g0[j], c0[k][j], a);
#pragma unroll
        for (int j = 0; j < 4; j++) a = __builtin_fmaf(g1[j], c1[k][j], a);
        d[k] = a;
      }
      const float lh = s_frac[2 * (pos * 9 + t)], lw = s_frac[2 * (pos * 9 + t) + 1];
      float gh = (1.f - lw) * (d[2] - d[0]) + lw * (d[3] - d[1]);
      float gw = (1.f - lh) * (d[1] - d[0]) + lh * (d[3] - d[2]);
      gh += __shfl_xor(gh, 1); gw += __shfl_xor(gw, 1);
      gh += __shfl_xor(gh, 2); gw += __shfl_xor(gw, 2);
      if (q == 0) {                              // one owner per (position, tap): plain read-modify-write
        s_goff[2 * (pos * 9 + t)] += gh;
        s_goff[2 * (pos * 9 + t) + 1] += gw;
      }
      if (!in) {                                 // input gradient of a sample that left the window: straight to memory
        const f32x4b w4 = s_w[pos * 9 + t];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float wk = w4[k];
          if (wk == 0.f) continue;
          const int yy = (int)tp.y + (k >> 1), xx = (int)tp.x + (k & 1);
          float* gp2 = grad_in + (((int64_t)b * C + cc * kFCh + q * 8) * H + yy) * W + xx;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            atomicAdd(gp2 + (int64_t)j * HW, wk * g0[j]);
            atomicAdd(gp2 + (int64_t)(4 + j) * HW, wk * g1[j]);
          }
        }
      }
    }
    // ---- input gradient: every (8-channel group, window pixel) sums its list; consecutive lanes = consecutive pixels.  (The
    // f16 kernel's form -- sums parked in LDS, 128-byte rows of an [S,H,W,C] accumulator -- measured SLOWER here: 2416 -> 2516 us
    // of kernel plus a 0.2 ms pass that adds the accumulator to gradInput.  This kernel waits for its f32 MFMA jobs, not for
    // its atomics, and two more barriers per chunk cost more than the better-shaped adds return.)
    for (int wi = tid; wi < ((S2A_BWD_ABL & 2) ? 0 : 4 * kFPix); wi += 512) {
      const int q = wi / kFPix, pix = wi % kFPix;
      const unsigned l0 = s_start[pix], l1 = s_start[pix + 1];
      if (l0 == l1) continue;
      f32x4b s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
      for (unsigned l = l0; l < l1; l += 4) {    // four entries per trip (entry -> row -> two row reads are dependent LDS round trips)
        unsigned ent[4];
        float wk[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned li = min(l + k, l1 - 1);
          ent[k] = s_list[li];
          wk[k] = l + k < l1 ? s_lw[li] : 0.f;
        }
        f32x4b g0[4], g1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float* gp = s_G + ent[k] * kFGRow + q * 8;
          g0[k] = *reinterpret_cast<const f32x4b*>(gp);
          g1[k] = *reinterpret_cast<const f32x4b*>(gp + 4);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
          for (int j = 0; j < 4; j++) { s0[j] = __builtin_fmaf(wk[k], g0[k][j], s0[j]); s1[j] = __builtin_fmaf(wk[k], g1[k][j], s1[j]); }
      }
      const int yy = oy + pix / kFPW, xx = ox + pix % kFPW;      // (pixels outside the image have no list: w = 0 there)
      float* gp2 = grad_in + (((int64_t)b * C + cc * kFCh + q * 8) * H + yy) * W + xx;
      if (S2A_BWD_ABL & 1) { if (s0[0] == 12345.f) gp2[0] = s0[1] + s0[2] + s0[3] + s1[0] + s1[1] + s1[2] + s1[3]; continue; }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        atomicAdd(gp2 + (int64_t)j * HW, s0[j]);
        atomicAdd(gp2 + (int64_t)(4 + j) * HW, s1[j]);
      }
    }
  }
  __syncthreads();
  // ---- offset gradient of the tile: [S, 18, H, W], channel 2 t = dy, 2 t + 1 = dx
  for (int i = tid; i < 18 * kFPos; i += 512) {
    const int ch = i / kFPos, pos = i % kFPos;
    const int y = ty0 + (pos >> 3), xq = tx0 + (pos & 7);
    if (y < H && xq < W)
      grad_off[((int64_t)b * 18 + ch) * HW + (int64_t)y * W + xq] = s_goff[2 * (pos * 9 + (ch >> 1)) + (ch & 1)];
  }
}

// ================================================================= fused weight gradient (f16, AlignConv geometry)
// deform_conv_backward_parameters_cuda (deform_conv_cuda.cpp:376-489): gradWeight[o, c, tap] = sum over positions of
// gradOutput[o, p] * columns[c, tap, p].  The reference (and the first version here) writes the sampled columns
// [C*9, N] to HBM (604 MB at P3, batch 8) and runs a GEMM over them.  Here the contraction index is the POSITION, so both
// MFMA operands would have to be read "down the rows" of their natural LDS images ([position][channel] columns as the
// forward's loaders make them, [position][out channel] gradOutput as it sits in NHWC memory): gfx950's transposing LDS
// read ds_read_b64_tr_b16 delivers exactly that -- four consecutive positions of one channel per lane -- so neither tile is
// ever transposed in software.
//   * a workgroup owns ONE 64-channel chunk and ONE row of three taps and a slice of the position tiles (split-K: 21 slices
//     x 12 owners = 252 workgroups on 256 CUs); its 3 x [O x 64] f32 results stay in registers over all its tiles
//     (8 waves x 6 accumulator tiles) and go out once, as a coalesced [O][3][64] block per workgroup that
//     k_dcn_bwd_weight_reduce sums over the slices (no atomics);
//   * per 4 x 16 position tile: gradOutput tile and the input patch of the chunk arrive in LDS (the NEXT tile's are already
//     in flight in registers), the eight waves blend the three taps' column tiles into LDS (bilinear corners from the patch,
//     as the forward does), then read both operands with transposing reads: 24 MFMAs per wave and tile.
constexpr int kWPos = 64;                       // positions per tile (4 x 16)
constexpr int kWgradMaxBlocks = 320;            // workgroups of a weight-gradient launch (<= CUs of the device): partial-result blocks
constexpr int kWColRow = 192;                   // bytes per position of a column tile: 64 halfs + pad (pitch = 48 dwords: the four
                                                // rows of a transposing read fall into four different 16-bank groups)
constexpr int kWPatchPix = kBPix;               // 12 x 24 window, 128 B per pixel (64 channels)
__device__ __forceinline__ int wgrad_go_pitch(int O) {   // bytes per position of the gradOutput tile: O halfs + pad to 16 (mod 64) dwords
  const int dw = O / 2;
  return (dw + ((16 - (dw & 63)) & 63)) * 4;
}
using s16x4b = __attribute__((ext_vector_type(4))) short;

__global__ __launch_bounds__(512, 2) void k_dcn_bwd_weight(const _Float16* __restrict__ x,        // NHWC [S,H,W,C]
                                                          const _Float16* __restrict__ go,       // NHWC [S,H,W,O]
                                                          const _Float16* __restrict__ offset,   // NCHW [S,18,H,W]
                                                          float* __restrict__ partial,           // [slice][owner][O][3 taps][64] f32
                                                          int S, int C, int H, int W, int O, int ksplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int gop = wgrad_go_pitch(O);
  char* s_go = smem;                                          // [64][gop]
  char* s_patch = s_go + kWPos * gop;                         // [288][128 B]
  char* s_col = s_patch + kWPatchPix * 128;                   // [3][64][kWColRow]
  BTap* s_tab = reinterpret_cast<BTap*>(s_col + 3 * kWPos * kWColRow);   // [3 * 64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int CC = C / 64;
  const int owner = blockIdx.x % (3 * CC), slice = blockIdx.x / (3 * CC);
  const int cc = owner / 3, ky = owner % 3;
  const int txn = (W + kBTW - 1) / kBTW, tyn = (H + kBTH - 1) / kBTH;
  const int ntiles = S * tyn * txn;
  const int64_t HW = (int64_t)H * W;
  const int OT = O / 32;                                      // out-channel tiles; wave w owns tile w (O <= 256)
  const bool mwave = wave < OT;
  f32x16b acc[3][2];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;

  // registers that carry the NEXT tile's gradOutput tile and patch (issued a tile ahead)
  constexpr int kGoVec = 5, kPaVec = 5;                       // 16-byte vectors per thread: 64 * 32 / 512 = 4 (O = 256), 288 * 8 / 512 = 4.5
  f16x8b gv[kGoVec], pvv[kPaVec];
  // Everything about a thread's vectors that does not depend on the tile is worked out once (position / channel group of
  // its gradOutput vectors, window row / column of its patch vectors, their LDS addresses), and the tile coordinates are
  // carried and advanced by ksplit tiles: the per-tile divisions (v / ovec with a run-time ovec, tile -> image / row / column)
  // were several hundred instructions per thread and tile in a kernel whose tile has 0.26 k cycles of MFMA.
  struct TilePos { int b, ty, tx; };             // image, tile row, tile column
  TilePos cur;
  {
    const int tile = min(slice, ntiles - 1), r = tile / txn;
    cur.tx = tile % txn; cur.ty = r % tyn; cur.b = r / tyn;
  }
  auto advance = [&](TilePos& p) {
    p.tx += ksplit;
    while (p.tx >= txn) {
      p.tx -= txn;
      if (++p.ty == tyn) { p.ty = 0; ++p.b; }
    }
  };
  const int ovec = O / 8;
  int g_yx[kGoVec], g_ch[kGoVec], g_lds[kGoVec], p_yx[kPaVec];     // y << 8 | x inside the tile / the window; -1: no such vector
#pragma unroll
  for (int i = 0; i < kGoVec; i++) {
    const int v = tid + 512 * i;
    const int pos = v / ovec, ch = v % ovec;
    g_yx[i] = v < kWPos * ovec ? ((pos >> 4) << 8 | (pos & 15)) : -1;
    g_ch[i] = ch * 8;
    g_lds[i] = pos * gop + ch * 16;
  }
#pragma unroll
  for (int i = 0; i < kPaVec; i++) {
    const int v = tid + 512 * i, pp = v >> 3;
    p_yx[i] = v < kWPatchPix * 8 ? ((pp / kBPW) << 8 | (pp % kBPW)) : -1;
  }
  // the table threads' two offsets, requested a tile ahead -- kept as loaded, each in a register of its own: a conversion (or
  // hipcc packing the two halves into one register) right behind the loads waits for EVERY request made before them
  // (vmcnt counts in order): that wait was the 2.9 k cycles per tile the stamps showed as "requests"
  unsigned off_y_raw = 0u, off_x_raw = 0u;
  auto issue = [&](const TilePos& tp_) {
    const int b = tp_.b, ty0 = tp_.ty * kBTH, tx0 = tp_.tx * kBTW;
    const int oy = ty0 - 3, ox = tx0 - 3;
    const _Float16* gb = go + (int64_t)b * HW * O;
    const _Float16* xb = x + (int64_t)b * HW * C + cc * 64 + (tid & 7) * 8;
#pragma unroll
    for (int i = 0; i < kGoVec; i++) {
      gv[i] = f16x8b{};
      const int y = ty0 + (g_yx[i] >> 8), xq = tx0 + (g_yx[i] & 255);
      if (g_yx[i] >= 0 && y < H && xq < W) gv[i] = *reinterpret_cast<const f16x8b*>(gb + ((int64_t)y * W + xq) * O + g_ch[i]);
    }
#pragma unroll
    for (int i = 0; i < kPaVec; i++) {
      pvv[i] = f16x8b{};
      const int yy = oy + (p_yx[i] >> 8), xx = ox + (p_yx[i] & 255);
      if (p_yx[i] >= 0 && yy >= 0 && yy < H && xx >= 0 && xx < W) pvv[i] = *reinterpret_cast<const f16x8b*>(xb + ((int64_t)yy * W + xx) * C);
    }
    if (tid < 3 * kWPos) {
      const int tl = tid / kWPos, pos = tid % kWPos, t = ky * 3 + tl;
      const int y = min(ty0 + (pos >> 4), H - 1), xq = min(tx0 + (pos & 15), W - 1);
      const _Float16* ob = offset + ((int64_t)b * 18) * HW + (int64_t)y * W + xq;
      off_y_raw = *reinterpret_cast<const unsigned short*>(ob + (int64_t)(2 * t) * HW);
      off_x_raw = *reinterpret_cast<const unsigned short*>(ob + (int64_t)(2 * t + 1) * HW);
    }
  };
  auto land = [&]() {
#pragma unroll
    for (int i = 0; i < kGoVec; i++)
      if (g_yx[i] >= 0) *reinterpret_cast<f16x8b*>(s_go + g_lds[i]) = gv[i];
#pragma unroll
    for (int i = 0; i < kPaVec; i++)
      if (p_yx[i] >= 0) *reinterpret_cast<f16x8b*>(s_patch + (tid + 512 * i) * 16) = pvv[i];
  };

#ifdef S2A_MEASURE_F16W
  unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c7 = 0, c4 = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
  int tile = slice;
  if (tile < ntiles) issue(cur);
  for (; tile < ntiles; tile += ksplit) {
    const int b = cur.b, ty0 = cur.ty * kBTH, tx0 = cur.tx * kBTW;
    const int oy = ty0 - 3, ox = tx0 - 3;
#ifdef S2A_MEASURE_F16W
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();                             // the previous tile's operands have been read
#ifdef S2A_MEASURE_F16W
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
    land();
    if (tid < 3 * kWPos) {                       // sampling table of this tile's three taps
      const int tl = tid / kWPos, pos = tid % kWPos, t = ky * 3 + tl;
      const int y = ty0 + (pos >> 4), xq = tx0 + (pos & 15);
      BTap tp;
      tp.y = 0; tp.x = 0; tp.flags = 0u;
      for (int k = 0; k < 4; k++) tp.w[k] = (_Float16)0.f;
      if (y < H && xq < W) {
        const float off_y = (float)__builtin_bit_cast(_Float16, (unsigned short)off_y_raw);
        const float off_x = (float)__builtin_bit_cast(_Float16, (unsigned short)off_x_raw);
        const float h_im = (float)(y - 1 + ky) + off_y, w_im = (float)(xq - 1 + tl) + off_x;
        if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const float lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;
          const bool t_ok = h_low >= 0, b_ok = h_low + 1 <= H - 1, l_ok = w_low >= 0, r_ok = w_low + 1 <= W - 1;
          tp.w[0] = (_Float16)((t_ok && l_ok) ? hh * hw : 0.f);
          tp.w[1] = (_Float16)((t_ok && r_ok) ? hh * lw : 0.f);
          tp.w[2] = (_Float16)((b_ok && l_ok) ? lh * hw : 0.f);
          tp.w[3] = (_Float16)((b_ok && r_ok) ? lh * lw : 0.f);
          tp.y = (short)h_low;
          tp.x = (short)w_low;
          const bool in = h_low >= oy && h_low + 1 <= oy + kBPH - 1 && w_low >= ox && w_low + 1 <= ox + kBPW - 1;
          const int py = min(max(h_low - oy, 0), kBPH - 2), px = min(max(w_low - ox, 0), kBPW - 2);
          tp.flags = 1u | (in ? 2u : 0u) | ((unsigned)(py * kBPW + px) << 2);
        }
      }
      s_tab[tid] = tp;
    }
    __syncthreads();
#ifdef S2A_MEASURE_F16W
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
#ifdef S2A_MEASURE_F16W
    const unsigned long long t2b = t2;
#endif
    // ---- column tiles of the three taps: item = (tap, position, 8-channel group), one tap per trip; the next tile's ten
    // requests per thread go out first.  (Stamps, `scripts/bwd32_stamps.sh f16`: a tile
    // is ~8.8 k cycles -- barrier 0.8, land + table 1.3, blend 2.8, MFMA 1.2 and ~2.5 k for the 69 KB of requests WHEREVER they are
    // placed, whether they hit cache or not, with 25 instructions around each or one: the kernel moves ~330 KB through LDS and
    // 69 KB into registers per tile and 0.26 k cycles of MFMA -- it is bound by that movement, not by the matrix pipe.)
    if (tile + ksplit < ntiles) {                // next tile's loads: in flight under the blend and the MFMAs
      advance(cur);
      issue(cur);
    }
#pragma unroll
    for (int trip = 0; trip < 3; trip++) {
      const int it = tid + 512 * trip;
      const int tl = it >> 9, r = it & 511, pos = r >> 3, q = r & 7;
      const BTap tp = s_tab[tl * kWPos + pos];
      char* dst = s_col + (tl * kWPos + pos) * kWColRow + q * 16;
      // A sample whose corners left the window (wild offsets) reads the input from memory.  That path must not share a basic
      // block join with the usual one: behind a join hipcc waits for ALL outstanding vector-memory operations -- the next
      // tile's requests included (their whole latency, 2.9 k cycles per tile, sat in front of the first blend item).
      if (!__any((tp.flags & 3u) == 1u)) {
        // usual case, LDS only (an invalid sample: all-zero weights, window pixel 0)
        const char* p0 = s_patch + ((int)(tp.flags >> 2) * 8 + q) * 16;
        const f16x8b c0 = *reinterpret_cast<const f16x8b*>(p0), c1 = *reinterpret_cast<const f16x8b*>(p0 + 128);
        const f16x8b c2 = *reinterpret_cast<const f16x8b*>(p0 + kBPW * 128), c3 = *reinterpret_cast<const f16x8b*>(p0 + kBPW * 128 + 128);
        const float w0 = (float)tp.w[0], w1 = (float)tp.w[1], w2 = (float)tp.w[2], w3 = (float)tp.w[3];
        f16x8b outv;
#pragma unroll
        for (int j = 0; j < 8; j++)
          outv[j] = (_Float16)(w0 * (float)c0[j] + w1 * (float)c1[j] + w2 * (float)c2[j] + w3 * (float)c3[j]);
        *reinterpret_cast<f16x8b*>(dst) = outv;
      } else {
        f16x8b outv = {};
        if (tp.flags & 1u) {
          f16x8b c4[4];
          if (tp.flags & 2u) {
            const char* p0 = s_patch + ((int)(tp.flags >> 2) * 8 + q) * 16;
            c4[0] = *reinterpret_cast<const f16x8b*>(p0);
            c4[1] = *reinterpret_cast<const f16x8b*>(p0 + 128);
            c4[2] = *reinterpret_cast<const f16x8b*>(p0 + kBPW * 128);
            c4[3] = *reinterpret_cast<const f16x8b*>(p0 + kBPW * 128 + 128);
          } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const int yy = min(max((int)tp.y + (k >> 1), 0), H - 1), xx = min(max((int)tp.x + (k & 1), 0), W - 1);
              c4[k] = *reinterpret_cast<const f16x8b*>(x + ((int64_t)b * HW + (int64_t)yy * W + xx) * C + cc * 64 + q * 8);
            }
          }
          const float w0 = (float)tp.w[0], w1 = (float)tp.w[1], w2 = (float)tp.w[2], w3 = (float)tp.w[3];
#pragma unroll
          for (int j = 0; j < 8; j++)
            outv[j] = (_Float16)(w0 * (float)c4[0][j] + w1 * (float)c4[1][j] + w2 * (float)c4[2][j] + w3 * (float)c4[3][j]);
        }
        *reinterpret_cast<f16x8b*>(dst) = outv;
      }
    }
    __syncthreads();
#ifdef S2A_MEASURE_F16W
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
    // ---- gradW tiles += gradOutput^T . columns over the 64 positions: both operands by transposing reads
    // (lane 16 g + 4 q + p supplies row q, elements 4 p .. 4 p + 3 of its group's 4 x 16 block and receives column
    // (lane & 15) of the four rows; group g covers rows 8 (g >> 1) + 4 r .. + 3 of the k-step and columns 16 (g & 1) .. + 15)
    if (mwave) {
      const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
      const char* a_base = s_go + (8 * (g >> 1) + qq) * gop + (wave * 32 + 16 * (g & 1) + 4 * pp) * 2;
      const char* b_base = s_col + (8 * (g >> 1) + qq) * kWColRow + (16 * (g & 1) + 4 * pp) * 2;
#pragma unroll
      for (int ks = 0; ks < kWPos / 16; ks++) {
        auto tr = [&](const char* p) {
          return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4b*)p);
        };
        auto frag = [&](const char* p, int pitch) {
          const s16x4b lo = tr(p + (ks * 16) * pitch), hi = tr(p + (ks * 16 + 4) * pitch);
          const __attribute__((ext_vector_type(8))) short v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          return __builtin_bit_cast(f16x8b, v);
        };
        const f16x8b A = frag(a_base, gop);
#pragma unroll
        for (int tl = 0; tl < 3; tl++)
#pragma unroll
          for (int ct = 0; ct < 2; ct++) {
            const f16x8b Bf = frag(b_base + tl * kWPos * kWColRow + ct * 64, kWColRow);
            acc[tl][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, Bf, acc[tl][ct], 0, 0, 0);
          }
      }
    }
#ifdef S2A_MEASURE_F16W
    const unsigned long long t4 = __builtin_amdgcn_s_memtime();
    c0 += t1 - t0; c1 += t2 - t1; c2 += t2b - t2; c3 += t3 - t2b; c7 += t4 - t3; c4++;
#endif
  }
#ifdef S2A_MEASURE_F16W
  if (tid == 0) {
    atomicAdd(&g_bwd_dbg[0], c0); atomicAdd(&g_bwd_dbg[1], c1); atomicAdd(&g_bwd_dbg[2], c2); atomicAdd(&g_bwd_dbg[3], c3);
    atomicAdd(&g_bwd_dbg[7], c7); atomicAdd(&g_bwd_dbg[4], c4); atomicAdd(&g_bwd_dbg[5], 1ull);
    atomicAdd(&g_bwd_dbg[6], __builtin_amdgcn_s_memtime() - t_begin);
  }
#endif
  // ---- results: rows = out channels (4 consecutive per register quad), columns = channels of the chunk; one block per workgroup
  if (mwave) {
    float* part = partial + (int64_t)blockIdx.x * O * 192;      // (blockIdx = slice * owners + owner)
#pragma unroll
    for (int tl = 0; tl < 3; tl++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int o = wave * 32 + 8 * (r >> 2) + (r & 3) + 4 * (lane >> 5);
          part[(o * 3 + tl) * 64 + ct * 32 + (lane & 31)] = acc[tl][ct][r];
        }
  }
}

// ================================================================= fused weight gradient (f32 tensors as three bf16 planes)
// The f32 matrix instruction runs at 1/16 of the 16-bit rate: k_dcn_bwd_weight_f32 below is its MFMA stream (1.1 of 1.58 ms at
// P3 x 8).  An f32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significand bits; hi = rne(x), mid = rne(x - hi),
// lo = rne(x - hi - mid)), and a product of two such sums, without the three terms below 2^-24 of it, is six bf16 products:
//     a b ~ a1 b1 + a1 b2 + a2 b1 + a1 b3 + a3 b1 + a2 b2        (f32 accumulation on the matrix cores, |error| <~ 3 * 2^-24 |a b|:
//                                                                  the size of ONE f32 rounding of the product)
// -- six 16-bit MFMAs for eight f32 ones of half their length: 2.7 x fewer matrix cycles at the accuracy of f32 arithmetic
// (the tests hold it to the same 1e-4 bound as the f32 instruction; measured difference to it ~1e-6 relative).
// Dataflow = k_dcn_bwd_weight's (f16) on 4 x 8 position tiles (window 12 x 16, as the f32 kernels): the gradOutput tile is
// split into its three planes when it lands in LDS, the input patch stays f32 (the bilinear blend runs in f32, as the
// reference's), every blended column value is split as it is written, and the MFMA waves read all planes with the
// transposing 16-bit LDS read.  One workgroup = one 64-channel chunk x one row of three taps x a slice of the tiles; partial
// blocks and the fixed-order reduce as before (bit-identical from run to run).  S2A_BWD_F32_WEIGHT=mfma32 selects the f32
// instruction's kernel (A/B and the tests' cross-check).
constexpr int kXPos = kFPos;                    // 32 positions per tile
constexpr int kXPatRow = 272;                   // bytes per window pixel of the f32 patch (64 floats + 16: conflict-light corner reads)
constexpr int kXTabRow = 32;                    // bytes per sampling-table entry
constexpr int kXColRow = 160;                   // bytes per position of a column plane: 64 bf16 + pad (pitch = 40 dwords: the four rows of a
                                                // transposing read start in banks 0, 40, 16, 56 -- four different 16-bank groups, as with the
                                                // f16 kernel's 48; 192 B would put the three planes of O = 256 past the LDS)
struct alignas(16) XTap {
  short y, x;        // top-left bilinear corner, image coordinates
  unsigned flags;    // bit 0: sample valid; bit 1: all four corners inside the LDS window; bits 31..2: window pixel index
  float w[4];        // hh*hw, hh*lw, lh*hw, lh*lw; 0 where the corner is outside the image
  unsigned pad[2];
};
static_assert(sizeof(XTap) == kXTabRow, "table entry");
__host__ __device__ inline int wgrad_x3_lds_bytes(int O) {
  const int dw = O / 2, gop = (dw + ((16 - (dw & 63)) & 63)) * 4;
  return 3 * kXPos * gop + kFPix * kXPatRow + 9 * kXPos * kXColRow + 3 * kXPos * kXTabRow;
}
using bf16x8b = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4b = __attribute__((ext_vector_type(4))) __bf16;

__global__ __launch_bounds__(512) void k_dcn_bwd_weight_x3(const float* __restrict__ x,        // NHWC [S,H,W,C]
                                                          const float* __restrict__ go,       // NHWC [S,H,W,O]
                                                          const float* __restrict__ offset,   // NCHW [S,18,H,W]
                                                          float* __restrict__ partial,        // [slice][owner][O][3 taps][64] f32
                                                          int S, int C, int H, int W, int O, int ksplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int gop = wgrad_go_pitch(O);
  char* s_go = smem;                                          // [3 planes][32][gop] bf16
  char* s_patch = s_go + 3 * kXPos * gop;                     // [192][kXPatRow] f32
  char* s_col = s_patch + kFPix * kXPatRow;                   // [3 planes][3 taps][32][kXColRow] bf16
  XTap* s_tab = reinterpret_cast<XTap*>(s_col + 9 * kXPos * kXColRow);   // [3 * 32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int CC = C / 64;
  const int owner = blockIdx.x % (3 * CC), slice = blockIdx.x / (3 * CC);
  const int cc = owner / 3, ky = owner % 3;
  const int txn = (W + kFTW - 1) / kFTW, tyn = (H + kFTH - 1) / kFTH;
  const int ntiles = S * tyn * txn;
  const int64_t HW = (int64_t)H * W;
  const int OT = O / 32;                                      // out-channel tiles; wave w owns tile w (O <= 256)
  const bool mwave = wave < OT;
  f32x16b acc[3][2];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][c][r] = 0.f;

  // registers that carry the NEXT tile's gradOutput tile and patch (issued a tile ahead): 32 x O / 4 and 192 x 16 vectors of 4 floats
  constexpr int kGoVec = 4, kPaVec = 6;
  f32x4b gv[kGoVec], pvv[kPaVec];
  struct TilePos { int b, ty, tx; };
  TilePos cur;
  {
    const int tile = min(slice, ntiles - 1), r = tile / txn;
    cur.tx = tile % txn; cur.ty = r % tyn; cur.b = r / tyn;
  }
  auto advance = [&](TilePos& p) {
    p.tx += ksplit;
    while (p.tx >= txn) {
      p.tx -= txn;
      if (++p.ty == tyn) { p.ty = 0; ++p.b; }
    }
  };
  const int ovec = O / 4;
  int g_yx[kGoVec], g_ch[kGoVec], g_lds[kGoVec], p_yx[kPaVec], p_lds[kPaVec];     // y << 8 | x inside the tile / the window; -1: no such vector
#pragma unroll
  for (int i = 0; i < kGoVec; i++) {
    const int v = tid + 512 * i;
    const int pos = v / ovec, ch = v % ovec;
    g_yx[i] = v < kXPos * ovec ? ((pos >> 3) << 8 | (pos & 7)) : -1;
    g_ch[i] = ch * 4;
    g_lds[i] = pos * gop + ch * 8;
  }
#pragma unroll
  for (int i = 0; i < kPaVec; i++) {
    const int v = tid + 512 * i, pp = v >> 4;                  // 16 vectors of 4 channels per pixel
    p_yx[i] = v < kFPix * 16 ? ((pp / kFPW) << 8 | (pp % kFPW)) : -1;
    p_lds[i] = pp * kXPatRow + (v & 15) * 16;
  }
  float off_y_raw = 0.f, off_x_raw = 0.f;        // the table threads' two offsets, requested a tile ahead
  auto issue = [&](const TilePos& tp_) {
    const int b = tp_.b, ty0 = tp_.ty * kFTH, tx0 = tp_.tx * kFTW;
    const int oy = ty0 - 3, ox = tx0 - 3;
    const float* gb = go + (int64_t)b * HW * O;
    const float* xb = x + (int64_t)b * HW * C + cc * 64 + (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < kGoVec; i++) {
      gv[i] = f32x4b{0.f, 0.f, 0.f, 0.f};
      const int y = ty0 + (g_yx[i] >> 8), xq = tx0 + (g_yx[i] & 255);
      if (g_yx[i] >= 0 && y < H && xq < W) gv[i] = *reinterpret_cast<const f32x4b*>(gb + ((int64_t)y * W + xq) * O + g_ch[i]);
    }
#pragma unroll
    for (int i = 0; i < kPaVec; i++) {
      pvv[i] = f32x4b{0.f, 0.f, 0.f, 0.f};
      const int yy = oy + (p_yx[i] >> 8), xx = ox + (p_yx[i] & 255);
      if (p_yx[i] >= 0 && yy >= 0 && yy < H && xx >= 0 && xx < W) pvv[i] = *reinterpret_cast<const f32x4b*>(xb + ((int64_t)yy * W + xx) * C);
    }
    if (tid < 3 * kXPos) {
      const int tl = tid / kXPos, pos = tid % kXPos, t = ky * 3 + tl;
      const int y = min(ty0 + (pos >> 3), H - 1), xq = min(tx0 + (pos & 7), W - 1);
      const float* ob = offset + ((int64_t)b * 18) * HW + (int64_t)y * W + xq;
      off_y_raw = ob[(int64_t)(2 * t) * HW];
      off_x_raw = ob[(int64_t)(2 * t + 1) * HW];
    }
  };
  auto land = [&]() {
#pragma unroll
    for (int i = 0; i < kGoVec; i++)
      if (g_yx[i] >= 0) {
        bf16x4b h, m, l;
#pragma unroll
        for (int j = 0; j < 4; j++) { __bf16 a, b2, c; split3(gv[i][j], a, b2, c); h[j] = a; m[j] = b2; l[j] = c; }
        *reinterpret_cast<bf16x4b*>(s_go + g_lds[i]) = h;
        *reinterpret_cast<bf16x4b*>(s_go + kXPos * gop + g_lds[i]) = m;
        *reinterpret_cast<bf16x4b*>(s_go + 2 * kXPos * gop + g_lds[i]) = l;
      }
#pragma unroll
    for (int i = 0; i < kPaVec; i++)
      if (p_yx[i] >= 0) *reinterpret_cast<f32x4b*>(s_patch + p_lds[i]) = pvv[i];
  };

  int tile = slice;
  if (tile < ntiles) issue(cur);
  for (; tile < ntiles; tile += ksplit) {
    const int b = cur.b, ty0 = cur.ty * kFTH, tx0 = cur.tx * kFTW;
    const int oy = ty0 - 3, ox = tx0 - 3;
    __syncthreads();                             // the previous tile's operands have been read
    land();
    if (tid < 3 * kXPos) {                       // sampling table of this tile's three taps
      const int tl = tid / kXPos, pos = tid % kXPos;
      const int y = ty0 + (pos >> 3), xq = tx0 + (pos & 7);
      XTap tp;
      tp.y = 0; tp.x = 0; tp.flags = 0u; tp.pad[0] = 0u; tp.pad[1] = 0u;
      for (int k = 0; k < 4; k++) tp.w[k] = 0.f;
      if (y < H && xq < W) {
        const float h_im = (float)(y - 1 + ky) + off_y_raw, w_im = (float)(xq - 1 + tl) + off_x_raw;
        if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const float lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;
          const bool t_ok = h_low >= 0, b_ok = h_low + 1 <= H - 1, l_ok = w_low >= 0, r_ok = w_low + 1 <= W - 1;
          tp.w[0] = (t_ok && l_ok) ? hh * hw : 0.f;
          tp.w[1] = (t_ok && r_ok) ? hh * lw : 0.f;
          tp.w[2] = (b_ok && l_ok) ? lh * hw : 0.f;
          tp.w[3] = (b_ok && r_ok) ? lh * lw : 0.f;
          tp.y = (short)h_low;
          tp.x = (short)w_low;
          const bool in = h_low >= oy && h_low + 1 <= oy + kFPH - 1 && w_low >= ox && w_low + 1 <= ox + kFPW - 1;
          const int py = min(max(h_low - oy, 0), kFPH - 2), px = min(max(w_low - ox, 0), kFPW - 2);
          tp.flags = 1u | (in ? 2u : 0u) | ((unsigned)(py * kFPW + px) << 2);
        }
      }
      s_tab[tid] = tp;
    }
    __syncthreads();
    if (tile + ksplit < ntiles) {                // next tile's loads: in flight under the blend and the MFMAs
      advance(cur);
      issue(cur);
    }
    // ---- column tiles of the three taps, blended in f32 (im2col_bilinear, kernel.cu:83-114: v1..v4 weighted in this order)
    // and split into the three planes: item = (tap, position, 8-channel group), 768 items.  (1 536 items of four channels, three
    // per thread, balance the trips but measured 1 023 -> 1 062 us: twice the table reads and address arithmetic.)
#pragma unroll
    for (int trip = 0; trip < 2; trip++) {
      const int it = tid + 512 * trip;
      if (it >= 3 * kXPos * 8) continue;
      const int tl = it >> 8, r = it & 255, pos = r >> 3, q = r & 7;
      const XTap tp = s_tab[tl * kXPos + pos];
      float v8[8];
#pragma unroll
      for (int j = 0; j < 8; j++) v8[j] = 0.f;
      // (as in the f16 kernel: the memory path of a sample that left the window must not share a join with the LDS path)
      if (!__any((tp.flags & 3u) == 1u)) {
        const char* p0 = s_patch + (int)(tp.flags >> 2) * kXPatRow + q * 32;
#pragma unroll
        for (int hv = 0; hv < 2; hv++) {
          const f32x4b c0 = *reinterpret_cast<const f32x4b*>(p0 + hv * 16), c1 = *reinterpret_cast<const f32x4b*>(p0 + kXPatRow + hv * 16);
          const f32x4b c2 = *reinterpret_cast<const f32x4b*>(p0 + kFPW * kXPatRow + hv * 16);
          const f32x4b c3 = *reinterpret_cast<const f32x4b*>(p0 + kFPW * kXPatRow + kXPatRow + hv * 16);
#pragma unroll
          for (int j = 0; j < 4; j++) v8[hv * 4 + j] = tp.w[0] * c0[j] + tp.w[1] * c1[j] + tp.w[2] * c2[j] + tp.w[3] * c3[j];
        }
      } else if (tp.flags & 1u) {
        f32x4b c4[4][2];
        if (tp.flags & 2u) {
          const char* p0 = s_patch + (int)(tp.flags >> 2) * kXPatRow + q * 32;
#pragma unroll
          for (int k = 0; k < 4; k++)
#pragma unroll
            for (int hv = 0; hv < 2; hv++)
              c4[k][hv] = *reinterpret_cast<const f32x4b*>(p0 + (k >> 1) * kFPW * kXPatRow + (k & 1) * kXPatRow + hv * 16);
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int yy = min(max((int)tp.y + (k >> 1), 0), H - 1), xx = min(max((int)tp.x + (k & 1), 0), W - 1);
            const float* gp = x + ((int64_t)b * HW + (int64_t)yy * W + xx) * C + cc * 64 + q * 8;
            c4[k][0] = *reinterpret_cast<const f32x4b*>(gp);
            c4[k][1] = *reinterpret_cast<const f32x4b*>(gp + 4);
          }
        }
#pragma unroll
        for (int hv = 0; hv < 2; hv++)
#pragma unroll
          for (int j = 0; j < 4; j++)
            v8[hv * 4 + j] = tp.w[0] * c4[0][hv][j] + tp.w[1] * c4[1][hv][j] + tp.w[2] * c4[2][hv][j] + tp.w[3] * c4[3][hv][j];
      }
      bf16x8b h, m, l;
#pragma unroll
      for (int j = 0; j < 8; j++) { __bf16 a, b2, c; split3(v8[j], a, b2, c); h[j] = a; m[j] = b2; l[j] = c; }
      char* dst = s_col + (tl * kXPos + pos) * kXColRow + q * 16;
      *reinterpret_cast<bf16x8b*>(dst) = h;
      *reinterpret_cast<bf16x8b*>(dst + 3 * kXPos * kXColRow) = m;
      *reinterpret_cast<bf16x8b*>(dst + 6 * kXPos * kXColRow) = l;
    }
    __syncthreads();
    // ---- gradW tiles += gradOutput^T . columns over the 32 positions, six plane products per tile; both operands by
    // transposing reads (lane maps: k_dcn_bwd_weight)
    if (mwave) {
      const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
      const char* a_base = s_go + (8 * (g >> 1) + qq) * gop + (wave * 32 + 16 * (g & 1) + 4 * pp) * 2;
      const char* b_base = s_col + (8 * (g >> 1) + qq) * kXColRow + (16 * (g & 1) + 4 * pp) * 2;
#pragma unroll
      for (int ks = 0; ks < kXPos / 16; ks++) {
        auto tr = [&](const char* p) {
          return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4b*)p);
        };
        auto frag = [&](const char* p, int pitch) {
          const s16x4b lo = tr(p + (ks * 16) * pitch), hi = tr(p + (ks * 16 + 4) * pitch);
          const __attribute__((ext_vector_type(8))) short v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          return __builtin_bit_cast(bf16x8b, v);
        };
        bf16x8b A[3];
#pragma unroll
        for (int pa = 0; pa < 3; pa++) A[pa] = frag(a_base + pa * kXPos * gop, gop);
#pragma unroll
        for (int tl = 0; tl < 3; tl++)
#pragma unroll
          for (int ct = 0; ct < 2; ct++) {
            bf16x8b Bf[3];
#pragma unroll
            for (int pb = 0; pb < 3; pb++) Bf[pb] = frag(b_base + (pb * 3 + tl) * kXPos * kXColRow + ct * 64, kXColRow);
            // small terms first
            acc[tl][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], Bf[0], acc[tl][ct], 0, 0, 0);
            acc[tl][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], Bf[2], acc[tl][ct], 0, 0, 0);
            acc[tl][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], Bf[1], acc[tl][ct], 0, 0, 0);
            acc[tl][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], Bf[0], acc[tl][ct], 0, 0, 0);
            acc[tl][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], Bf[1], acc[tl][ct], 0, 0, 0);
            acc[tl][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], Bf[0], acc[tl][ct], 0, 0, 0);
          }
      }
    }
  }
  // ---- results: rows = out channels (4 consecutive per register quad), columns = channels of the chunk; one block per workgroup
  if (mwave) {
    float* part = partial + (int64_t)blockIdx.x * O * 192;      // (blockIdx = slice * owners + owner)
#pragma unroll
    for (int tl = 0; tl < 3; tl++)
#pragma unroll
      for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int o = wave * 32 + 8 * (r >> 2) + (r & 3) + 4 * (lane >> 5);
          part[(o * 3 + tl) * 64 + ct * 32 + (lane & 31)] = acc[tl][ct][r];
        }
  }
}

// gradWeight += scale * (sum over the position slices of the workgroups' partial results).  The fused weight-gradient kernels
// used to add their 3 x [O x 64] register tiles into gradWeight with f32 atomics: 12.4 M lane atomics per call, 36 bytes apart
// (nine taps between two channels), 21 slices contending for every address -- 2.3 k cycles per tile of the f32 kernel, ~0.2 ms
// per call.  Now every workgroup stores its block once, coalesced, and ONE thread per element sums the slices in a fixed order:
// no atomics, and the gradient is bit-identical from run to run.
__global__ __launch_bounds__(256) void k_dcn_bwd_weight_reduce(const float* __restrict__ partial, float* __restrict__ grad_w,
                                                              float scale, int O, int C, int ksplit) {
  const int64_t block = (int64_t)O * 192;
  const int owners = 3 * (C / 64);
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= block * owners) return;
  const int owner = (int)(e / block);
  const int rem = (int)(e % block);
  const int o = rem / 192, tl = (rem / 64) % 3, c = rem & 63;
  const int cc = owner / 3, ky = owner % 3;
  float sum = 0.f;
  for (int sl = 0; sl < ksplit; sl++) sum += partial[((int64_t)sl * owners + owner) * block + rem];
  float* dst = grad_w + ((int64_t)o * C + cc * 64 + c) * 9 + ky * 3 + tl;
  *dst += scale * sum;
}

// ================================================================= fused weight gradient (f32, AlignConv geometry)
// k_dcn_bwd_weight's dataflow for float32 tensors on v_mfma_f32_16x16x4_f32 (64 FLOP / clk / SIMD; the 155 GFLOP of a P3 x 8
// call are 0.99 ms at the 2.4 GHz peak, 1.2 ms at the ~2.0 GHz the chip holds under this kernel).  With f32 operands nothing
// has to be transposed: the contraction index (the position) is the ROW of the column tiles in LDS, and an f32 MFMA operand
// is one value per lane -- lane (l & 15, l >> 4) reads element l & 15 of row 4 ks + (l >> 4): 16 consecutive floats per
// quarter wave, the four rows a pitch apart that is 16 (mod 64) floats, so the 64 lanes cover the 64 banks.
// A workgroup owns ONE 64-channel chunk and ONE row of three taps and a slice of the 4 x 8 position tiles (12 owners x 21
// slices = 252 workgroups); its 3 x [O x 64] f32 results stay in registers over all its tiles and go out once, as a block
// that k_dcn_bwd_weight_reduce sums over the slices and adds, scaled, to gradWeight (deform_conv_cuda.cpp:455-459).
// Twelve waves in two roles:
//   * waves 0-7 (two per SIMD) own out channels 32 w .. 32 w + 31 and only run MFMAs: 8 k-steps x 24 MFMAs (16x16x4) per
//     tile, the column operand from LDS (requested one k-step ahead), the gradOutput operand straight from memory -- in NHWC
//     the 32 out channels of one position are 128 contiguous bytes, and a wave needs just 16 values per lane and tile,
//     so the whole next tile's operand is requested between this tile's MFMAs and waits in registers (no LDS copy of
//     gradOutput at all; every element is loaded once per workgroup);
//   * waves 8-11 (one per SIMD) prepare the NEXT tile under them: input patch -> LDS (requested a tile ahead into registers),
//     sampling table, the three taps' column tiles blended into the other half of a double buffer.
// The first form of this kernel did everything with eight waves in sequence (load, blend, MFMA): 19.6 k cycles per tile for
// 12.3 k of MFMA -- the 80 KB a tile needs pass the CU's load path at ~32 B/clk (2.5 k cycles) whether they are requested in
// one burst or between the MFMAs of the same waves (a wave whose load waits for a queue slot issues no MFMA either).
// Two workgroup barriers per tile: B1 (buffers swap) and B2 in the middle of the MFMA phase, which separates the loaders'
// patch / table writes from their blend; the MFMA waves always arrive last, so only loaders wait.
constexpr int kFWColRow = 80;                  // floats per position of a column tile: 64 channels + 16 (the four k rows of an
                                               // operand read -- 16 consecutive floats each -- fall into four different 16-bank groups)
constexpr int kFWPatRow = 68;                  // floats per window pixel: 64 channels + 4
constexpr int kFWThreads = 768, kFWLoaders = 256;
constexpr int kFWColBuf = 3 * kFPos * kFWColRow;              // floats per column buffer
__host__ __device__ constexpr int wgrad_f32_lds_bytes() {
  return kFPix * kFWPatRow * 4 + 2 * kFWColBuf * 4 + 3 * kFPos * 16 + 3 * kFPos * 8 + 64;
}
__global__ __launch_bounds__(kFWThreads) void k_dcn_bwd_weight_f32(const float* __restrict__ x,        // NHWC [S,H,W,C]
                                                                  const float* __restrict__ go,       // NHWC [S,H,W,O]
                                                                  const float* __restrict__ offset,   // NCHW [S,18,H,W]
                                                                  float* __restrict__ partial,        // [slice][owner][O][3 taps][64]
                                                                  int S, int C, int H, int W, int O, int ksplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_patch = reinterpret_cast<float*>(smem);                           // [192][kFWPatRow]
  float* s_col = s_patch + kFPix * kFWPatRow;                                // [2][3][32][kFWColRow]
  f32x4b* s_w = reinterpret_cast<f32x4b*>(s_col + 2 * kFWColBuf);            // [3 * 32] corner weights
  FTap* s_tab = reinterpret_cast<FTap*>(s_w + 3 * kFPos);                    // [3 * 32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 8;
  const int ltid = tid - 512;                                                // loaders: 0 .. 255
  const int CC = C / 64;
  // workgroups are dealt to the eight XCDs round-robin (blockIdx % 8), each with its own L2: the twelve owners of a slice walk the
  // same tiles at the same pace (one gradOutput tile, four input patches between them), so they are numbered to sit on ONE XCD
  // -- logical index = XCD * (grid / 8) + blockIdx / 8 (the grid is a multiple of 8; indices past the last workgroup leave)
  const int logical = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  if (logical >= 3 * CC * ksplit) return;
  const int owner = logical % (3 * CC), slice = logical / (3 * CC);
  const int cc = owner / 3, ky = owner % 3;
  const int txn = (W + kFTW - 1) / kFTW, tyn = (H + kFTH - 1) / kFTH;
  const int ntiles = S * tyn * txn;
  const int nt = slice < ntiles ? (ntiles - slice + ksplit - 1) / ksplit : 0;     // tiles of this workgroup: slice + j * ksplit
  const int64_t HW = (int64_t)H * W;
  const bool mwave = wave < O / 32;                                          // MFMA wave w owns out-channel tile w (O <= 256)
  // tile coordinates are carried and advanced by ksplit tiles (a walk of a few steps): the integer divisions of tile -> (image,
  // row, column) cost ~100 instructions, per tile and wave, in streams that have no issue slots to spare beside the MFMAs
  struct TilePos { int b, ty, tx; };             // image, tile row, tile column
  TilePos first;
  {
    const int tile = min(slice, ntiles - 1), r = tile / txn;
    first.tx = tile % txn; first.ty = r % tyn; first.b = r / tyn;
  }
  auto advance = [&](TilePos& p) {
    p.tx += ksplit;
    while (p.tx >= txn) {
      p.tx -= txn;
      if (++p.ty == tyn) { p.ty = 0; ++p.b; }
    }
  };
#ifdef S2A_MEASURE
  unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c7 = 0, c8 = 0, c9 = 0, c10 = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
  if (nt == 0) return;

  if (loader) {
    // =========================================================== loader waves
    // VALU issue on a SIMD goes by priority, then age: as the youngest of three waves, beside two that always have an MFMA
    // pending, a loader got ONE vector instruction per MFMA slot (64 cycles: 5.4 k cycles for the twelve stores of a patch)
    __builtin_amdgcn_s_setprio(3);
    constexpr int kPaVec = 12;                   // 16-byte vectors per thread: 192 pixels x 16 / 256
    f32x4b pvv[kPaVec];
    unsigned okm = 0u;
    float off_yx[2] = {0.f, 0.f};
    // branch-free requests: an element outside the image loads from a clamped address and is zeroed when it lands
    // vector i of a thread = window row i, pixel ltid >> 4 of that row, channels 4 (ltid & 15) ..: the column part of the address is
    // per tile, the row part one multiply-add per request (issue slots are what a loader is short of beside two MFMA streams)
    static_assert(kFPW == 16 && kFPH == kPaVec && kFWLoaders == 256, "loader geometry");
    const int lpx = ltid >> 4, lq = ltid & 15;
    auto request = [&](const TilePos& tp_) {
      const int b = tp_.b, ty0 = tp_.ty * kFTH, tx0 = tp_.tx * kFTW;
      const int oy = ty0 - 3, xx = tx0 - 3 + lpx;
      const bool okx = xx >= 0 && xx < W;
      const float* colp = x + ((int64_t)b * HW + min(max(xx, 0), W - 1)) * C + cc * 64 + lq * 4;
      const int64_t rowpitch = (int64_t)W * C;
      okm = 0u;
#pragma unroll
      for (int i = 0; i < kPaVec; i++) {
        const int yy = oy + i;
        pvv[i] = *reinterpret_cast<const f32x4b*>(colp + min(max(yy, 0), H - 1) * rowpitch);
        okm |= ((okx && yy >= 0 && yy < H) ? 1u : 0u) << i;
      }
      const int e = min(ltid, 3 * kFPos - 1), tl = e / kFPos, pos = e % kFPos, t = ky * 3 + tl;
      const int y = min(ty0 + (pos >> 3), H - 1), xq = min(tx0 + (pos & 7), W - 1);
      const float* ob = offset + ((int64_t)b * 18) * HW + (int64_t)y * W + xq;
      off_yx[0] = ob[(int64_t)(2 * t) * HW];
      off_yx[1] = ob[(int64_t)(2 * t + 1) * HW];
    };
    TilePos cur = first;
    request(cur);
    for (int j = 0; j <= nt; j++) {
      const int b = cur.b, ty0 = cur.ty * kFTH, tx0 = cur.tx * kFTW;
      const int oy = ty0 - 3, ox = tx0 - 3;
      __syncthreads();                           // B1: the blend of tile j - 1 has read the patch and the table
      BWD_T(t1);
#ifdef S2A_MEASURE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t1a = __builtin_amdgcn_s_memtime();
      unsigned long long t1b = t1a;
#endif
      if (j < nt) {
        float* pw = s_patch + lpx * kFWPatRow + lq * 4;            // row i: + i * 16 pixels
        if (okm == (1u << kPaVec) - 1u) {
#pragma unroll
          for (int i = 0; i < kPaVec; i++) *reinterpret_cast<f32x4b*>(pw + i * kFPW * kFWPatRow) = pvv[i];
        } else {
          const f32x4b zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < kPaVec; i++) *reinterpret_cast<f32x4b*>(pw + i * kFPW * kFWPatRow) = ((okm >> i) & 1u) ? pvv[i] : zero;
        }
#ifdef S2A_MEASURE
        t1b = __builtin_amdgcn_s_memtime();
#endif
        if (ltid < 3 * kFPos) {                  // sampling table of this tile's three taps
          const int tl = ltid / kFPos, pos = ltid % kFPos;
          const int y = ty0 + (pos >> 3), xq = tx0 + (pos & 7);
          FTap tp;
          tp.y = 0; tp.x = 0; tp.flags = 0u;
          f32x4b w4 = {0.f, 0.f, 0.f, 0.f};
          if (y < H && xq < W) {
            const float h_im = (float)(y - 1 + ky) + off_yx[0], w_im = (float)(xq - 1 + tl) + off_yx[1];
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
              const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
              const float lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;
              const bool t_ok = h_low >= 0, b_ok = h_low + 1 <= H - 1, l_ok = w_low >= 0, r_ok = w_low + 1 <= W - 1;
              w4[0] = (t_ok && l_ok) ? hh * hw : 0.f;
              w4[1] = (t_ok && r_ok) ? hh * lw : 0.f;
              w4[2] = (b_ok && l_ok) ? lh * hw : 0.f;
              w4[3] = (b_ok && r_ok) ? lh * lw : 0.f;
              tp.y = (short)h_low;
              tp.x = (short)w_low;
              const bool in = h_low >= oy && h_low + 1 <= oy + kFPH - 1 && w_low >= ox && w_low + 1 <= ox + kFPW - 1;
              const int py = min(max(h_low - oy, 0), kFPH - 2), px = min(max(w_low - ox, 0), kFPW - 2);
              tp.flags = 1u | (in ? 2u : 0u) | ((unsigned)(py * kFPW + px) << 2);
            }
          }
          s_tab[ltid] = tp;
          s_w[ltid] = w4;
        }
      }
      BWD_T(t2);
      __syncthreads();                           // B2: patch and table of tile j are in LDS
      BWD_T(t3);
      if (j < nt) {
        if (j + 1 < nt) {                          // in flight under the blend and the next tile's first half
          advance(cur);
          if (!(S2A_BWD_ABL & 128)) request(cur);
        }
#ifdef S2A_MEASURE
        c10 += __builtin_amdgcn_s_memtime() - t3;
#endif
        // ---- column tiles of the three taps into buffer j & 1: item = (tap, position, 8-channel group)
        float* colb = s_col + (j & 1) * kFWColBuf;
        const int pos = ltid >> 3, q = ltid & 7;       // item (tap tl, position, 8-channel group): a thread's three items differ in the tap
        FTap tp[3];
        f32x4b w4[3];
        bool far = false;                              // a valid sample whose corners left the window (wild offsets): global reads
#pragma unroll
        for (int tl = 0; tl < 3; tl++) {
          tp[tl] = s_tab[tl * kFPos + pos];
          w4[tl] = s_w[tl * kFPos + pos];              // (all zero for an invalid sample)
          far |= (tp[tl].flags & 3u) == 1u;
        }
        if (S2A_BWD_ABL & 64) {
        } else if (!__any(far)) {
          // the usual case, branch-free: twenty-four corner reads in flight, then the arithmetic (an invalid sample reads pixel 0
          // with zero weights)
          // (taps 0 and 1 together, then tap 2: all three at once spill beside the twelve patch vectors already requested)
          auto fast = [&](auto T0, auto T1) {
            constexpr int t0 = decltype(T0)::value, t1 = decltype(T1)::value;
            f32x4b c0[t1 - t0][4], c1[t1 - t0][4];
#pragma unroll
            for (int tl = t0; tl < t1; tl++) {
              const float* p0 = s_patch + (int)(tp[tl].flags >> 2) * kFWPatRow + q * 8;
#pragma unroll
              for (int k = 0; k < 4; k++) {
                const float* pk = p0 + ((k >> 1) * kFPW + (k & 1)) * kFWPatRow;
                c0[tl - t0][k] = *reinterpret_cast<const f32x4b*>(pk);
                c1[tl - t0][k] = *reinterpret_cast<const f32x4b*>(pk + 4);
              }
            }
#pragma unroll
            for (int tl = t0; tl < t1; tl++) {
              f32x4b o0, o1;
              // deformable_im2col_bilinear (:110-112): w1 v1 + w2 v2 + w3 v3 + w4 v4, in that order
#pragma unroll
              for (int jj = 0; jj < 4; jj++) {
                o0[jj] = w4[tl][0] * c0[tl - t0][0][jj] + w4[tl][1] * c0[tl - t0][1][jj] + w4[tl][2] * c0[tl - t0][2][jj] + w4[tl][3] * c0[tl - t0][3][jj];
                o1[jj] = w4[tl][0] * c1[tl - t0][0][jj] + w4[tl][1] * c1[tl - t0][1][jj] + w4[tl][2] * c1[tl - t0][2][jj] + w4[tl][3] * c1[tl - t0][3][jj];
              }
              float* cp = colb + (tl * kFPos + pos) * kFWColRow + q * 8;
              *reinterpret_cast<f32x4b*>(cp) = o0;
              *reinterpret_cast<f32x4b*>(cp + 4) = o1;
            }
          };
          fast(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
          fast(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
        } else {
          for (int tl = 0; tl < 3; tl++) {
            f32x4b o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
            if (tp[tl].flags & 1u) {
              f32x4b c0[4], c1[4];
              if (tp[tl].flags & 2u) {
                const float* p0 = s_patch + (int)(tp[tl].flags >> 2) * kFWPatRow + q * 8;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                  const float* pk = p0 + ((k >> 1) * kFPW + (k & 1)) * kFWPatRow;
                  c0[k] = *reinterpret_cast<const f32x4b*>(pk);
                  c1[k] = *reinterpret_cast<const f32x4b*>(pk + 4);
                }
              } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                  const int yy = (int)tp[tl].y + (k >> 1), xx = (int)tp[tl].x + (k & 1);
                  c0[k] = f32x4b{0.f, 0.f, 0.f, 0.f};
                  c1[k] = c0[k];
                  if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const float* pk = x + ((int64_t)b * HW + (int64_t)yy * W + xx) * C + cc * 64 + q * 8;
                    c0[k] = *reinterpret_cast<const f32x4b*>(pk);
                    c1[k] = *reinterpret_cast<const f32x4b*>(pk + 4);
                  }
                }
              }
#pragma unroll
              for (int jj = 0; jj < 4; jj++) {
                o0[jj] = w4[tl][0] * c0[0][jj] + w4[tl][1] * c0[1][jj] + w4[tl][2] * c0[2][jj] + w4[tl][3] * c0[3][jj];
                o1[jj] = w4[tl][0] * c1[0][jj] + w4[tl][1] * c1[1][jj] + w4[tl][2] * c1[2][jj] + w4[tl][3] * c1[3][jj];
              }
            }
            float* cp = colb + (tl * kFPos + pos) * kFWColRow + q * 8;
            *reinterpret_cast<f32x4b*>(cp) = o0;
            *reinterpret_cast<f32x4b*>(cp + 4) = o1;
          }
        }
      }
#ifdef S2A_MEASURE
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      c3 += t2 - t1; c7 += t4 - t3; c8 += t1a - t1; c9 += t1b - t1a;
#endif
    }
#ifdef S2A_MEASURE
    if (ltid == 0) {
      atomicAdd(&g_bwd_dbg[3], c3); atomicAdd(&g_bwd_dbg[7], c7); atomicAdd(&g_bwd_dbg[8], c8); atomicAdd(&g_bwd_dbg[9], c9);
      atomicAdd(&g_bwd_dbg[10], c10);
    }
#endif
    return;
  }

  // =========================================================== MFMA waves
  // v_mfma_f32_16x16x4_f32, not 32x32x2: the same 64 FLOP / clk, but an accumulator tile is read and written once per FOUR
  // positions instead of once per two.  Back-to-back 32x32x2 MFMAs keep the SIMD's register-file write port busy with their
  // own 16-register results (16 x 4 of every 64 cycles), and everything else on that SIMD that returns a value to a register
  // -- the partner's operand reads, the loader's patch requests -- queues behind them (stamps: the loaders' twelve requests took
  // 7.0 k cycles per tile beside the MFMAs and 1.4 k with the MFMAs compiled out).
  // Wave w owns out channels 32 w .. 32 w + 31 as two 16-row tiles of interleaved channels (tile h, row i = channel 2 i + h:
  // one 8-byte request fetches both), and 3 taps x four 16-channel column tiles: 24 accumulators of 4 registers.
  f32x4b acc[2][12];
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int n = 0; n < 12; n++) acc[h][n] = f32x4b{0.f, 0.f, 0.f, 0.f};
  // gradOutput operand of a whole tile: k-step ks = positions 4 ks .. 4 ks + 3, lane (l & 15, l >> 4) holds the channel pair
  // 2 (l & 15), + 1 of position 4 ks + (l >> 4) = tile row ks >> 1, tile column 4 (ks & 1) + (l >> 4).  Address arithmetic
  // per tile, not per request: two row pointers... four (uniform) and two lane offsets; positions outside the image load from a
  // clamped address and count as zero.
  using f32x2b = __attribute__((ext_vector_type(2))) float;
  constexpr int kSteps = kFPos / 4;              // 8
  f32x2b a_cur[kSteps], a_nxt[kSteps];
  unsigned ok_cur = 0u, ok_nxt = 0u;
  const float* a_row[4];
  int a_col[2];
  const int a_lane = min(wave, O / 32 - 1) * 32 + 2 * (lane & 15), a_h = lane >> 4;
  auto setup_a = [&](const TilePos& tp_, unsigned& okbits) {
    const int b = tp_.b, ty0 = tp_.ty * kFTH, tx0 = tp_.tx * kFTW;
    unsigned oky = 0u, okx = 0u;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      a_row[r] = go + ((int64_t)b * HW + (int64_t)min(ty0 + r, H - 1) * W) * O;
      oky |= (ty0 + r < H ? 1u : 0u) << r;
    }
#pragma unroll
    for (int p2 = 0; p2 < 2; p2++) {
      const int xq = tx0 + 4 * p2 + a_h;
      a_col[p2] = min(xq, W - 1) * O + a_lane;
      okx |= (xq < W ? 1u : 0u) << p2;
    }
    okbits = 0u;
#pragma unroll
    for (int ks = 0; ks < kSteps; ks++) okbits |= (((oky >> (ks >> 1)) & (okx >> (ks & 1))) & 1u) << ks;
  };
  auto request_a = [&](int ks) { return *reinterpret_cast<const f32x2b*>(a_row[ks >> 1] + a_col[ks & 1]); };
  TilePos apos = first;
  setup_a(apos, ok_cur);
#pragma unroll
  for (int ks = 0; ks < kSteps; ks++) a_cur[ks] = request_a(ks);
  const float* b_lane = s_col + (lane >> 4) * kFWColRow + (lane & 15);
  for (int j = 0; j <= nt; j++) {
    BWD_T(t0);
    __syncthreads();                             // B1: column buffer (j - 1) & 1 holds tile j - 1
    BWD_T(t1);
    const float* b_base = b_lane + ((j - 1) & 1) * kFWColBuf;
    // the twelve column values of k-step ks + 1 are requested before the 24 MFMAs of k-step ks (order pinned: left alone,
    // hipcc reads each operand right in front of its MFMAs and waits out the LDS round trip every time)
    float Bv[2][12];
    auto fetch = [&](int ks, float (&bv)[12]) {
#pragma unroll
      for (int tl = 0; tl < 3; tl++)
#pragma unroll
        for (int ct = 0; ct < 4; ct++) bv[tl * 4 + ct] = b_base[(tl * kFPos + 4 * ks) * kFWColRow + ct * 16];
    };
    auto half = [&](int hh) {
#pragma unroll
      for (int ks = 4 * hh; ks < 4 * hh + 4; ks++) {
        if (ks + 1 < kSteps) fetch(ks + 1, Bv[(ks + 1) & 1]);
        // the next tile's operand: two requests per k-step of the FIRST half -- the registers are handed over at the end of the
        // tile, and a request made in the last k-steps would have its whole memory latency waited out there
        if (!(S2A_BWD_ABL & 128) && hh == 0) {
          a_nxt[2 * ks] = request_a(2 * ks);
          a_nxt[2 * ks + 1] = request_a(2 * ks + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool ok = (ok_cur >> ks) & 1u;
        const float A0 = ok ? a_cur[ks][0] : 0.f, A1 = ok ? a_cur[ks][1] : 0.f;
#pragma unroll
        for (int n = 0; n < 12; n++)
          if (!(S2A_BWD_ABL & 32)) {
            acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0, Bv[ks & 1][n], acc[0][n], 0, 0, 0);
            acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1, Bv[ks & 1][n], acc[1][n], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    const bool work = j >= 1 && mwave;
    if (work) {
      fetch(0, Bv[0]);
      if (j < nt) advance(apos);                 // (after the last tile: that tile again -- harmless reads)
      setup_a(apos, ok_nxt);
      half(0);
    }
    BWD_T(t2);
    __syncthreads();                             // B2 (the loaders' patch / table hand-over; nothing of this role depends on it)
    BWD_T(t3);
    if (work) {
      half(1);
#pragma unroll
      for (int ks = 0; ks < kSteps; ks++) a_cur[ks] = a_nxt[ks];
      ok_cur = ok_nxt;
    }
#ifdef S2A_MEASURE
    const unsigned long long t4 = __builtin_amdgcn_s_memtime();
    c1 += t1 - t0; c0 += (t2 - t1) + (t4 - t3); c2 += t3 - t2;
#endif
  }
  // ---- results: accumulator (h, tl * 4 + ct), register r, lane l = out channel 32 w + 2 (4 (l >> 4) + r) + h, channel 16 ct + (l & 15)
  // one block per workgroup, summed over the slices by k_dcn_bwd_weight_reduce
  if (mwave) {
    float* part = partial + (int64_t)logical * O * 192;         // (logical = slice * owners + owner)
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int n = 0; n < 12; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int o = wave * 32 + 2 * (4 * (lane >> 4) + r) + h;
          part[(o * 3 + (n >> 2)) * 64 + (n & 3) * 16 + (lane & 15)] = acc[h][n][r];
        }
  }
#ifdef S2A_MEASURE
  if (tid == 256) { atomicAdd(&g_bwd_dbg[11], c0); atomicAdd(&g_bwd_dbg[12], c1); atomicAdd(&g_bwd_dbg[13], c2); }
  if (tid == 0) {
    atomicAdd(&g_bwd_dbg[0], c0); atomicAdd(&g_bwd_dbg[1], c1); atomicAdd(&g_bwd_dbg[2], c2);
    atomicAdd(&g_bwd_dbg[4], (unsigned long long)nt); atomicAdd(&g_bwd_dbg[5], 1ull); atomicAdd(&g_bwd_dbg[6], __builtin_amdgcn_s_memtime() - t_begin);
  }
#endif
}

int make_geom(const s2a_dcn_params* pp, BwdGeom* g, const char* who) {
  S2A_CHECK_ARG(pp != nullptr, "%s: NULL params", who);
  const s2a_dcn_params& p = *pp;
  S2A_CHECK_ARG(p.kW > 0 && p.kH > 0 && p.dW > 0 && p.dH > 0 && p.dilationW > 0 && p.dilationH > 0 &&
                p.deformable_group > 0, "%s: bad kernel geometry", who);
  S2A_CHECK_ARG(p.batch >= 0 && p.channels > 0 && p.height > 0 && p.width > 0, "%s: bad shape", who);
  S2A_CHECK_ARG(p.channels % p.deformable_group == 0, "input channels must divide deformable group size");
  S2A_CHECK_ARG(p.dtype == S2A_DTYPE_F32 || p.dtype == S2A_DTYPE_F16 || p.dtype == S2A_DTYPE_F64, "%s: dtype", who);
  S2A_CHECK_ARG(p.layout == S2A_LAYOUT_NCHW, "%s: NCHW only", who);
  const int64_t Ho = (p.height + 2 * p.padH - (p.dilationH * (p.kH - 1) + 1)) / p.dH + 1;
  const int64_t Wo = (p.width + 2 * p.padW - (p.dilationW * (p.kW - 1) + 1)) / p.dW + 1;
  S2A_CHECK_ARG(Ho >= 1 && Wo >= 1, "%s: output size is too small", who);
  S2A_CHECK_ARG(p.height < (1 << 15) && p.width < (1 << 15) && p.channels * p.kH * p.kW < (1ll << 31),
                "%s: shape too large", who);
  *g = BwdGeom{(int)p.channels, (int)p.height, (int)p.width, p.kH, p.kW, p.padH, p.padW, p.dH, p.dW,
               p.dilationH, p.dilationW, p.deformable_group, (int)p.batch, (int)Ho, (int)Wo};
  return S2A_OK;
}

inline unsigned grid_for(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 1 << 20); }

}  // namespace

int build_flags_dcn_bwd() {
  int f = S2A_BWD_ABL ? 1 : 0;
#ifdef S2A_MEASURE
  f |= 2;
#endif
  return f;
}
}  // namespace s2a

#ifdef S2A_MEASURE
// measurement builds: the phase cycles of k_dcn_bwd_weight_f32 (wave 0 of every workgroup, summed) since the last call
extern "C" int s2a_debug_bwd_stamps(unsigned long long* host_dst) {
  S2A_HIP(hipDeviceSynchronize());
  S2A_HIP(hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(s2a::g_bwd_dbg), sizeof(unsigned long long) * 16));
  unsigned long long z[16] = {};
  S2A_HIP(hipMemcpyToSymbol(HIP_SYMBOL(s2a::g_bwd_dbg), z, sizeof(z)));
  return S2A_OK;
}
#endif

using namespace s2a;

extern "C" int s2a_deformable_im2col(const void* im, const void* offset, void* columns, const s2a_dcn_params* p,
                                     s2a_stream_t stream) {
  BwdGeom g;
  int rc = make_geom(p, &g, "deformable_im2col");
  if (rc != S2A_OK) return rc;
  const int64_t n = (int64_t)g.C * g.S * g.Ho * g.Wo;
  if (n == 0) return S2A_OK;
  S2A_CHECK_ARG(im && offset && columns, "deformable_im2col: NULL tensor");
  hipStream_t st = as_stream(stream);
  if (p->dtype == S2A_DTYPE_F64)
    k_def_im2col<double><<<grid_for(n), 256, 0, st>>>(n, (const double*)im, (const double*)offset, g, (double*)columns);
  else if (p->dtype == S2A_DTYPE_F32)
    k_def_im2col<float><<<grid_for(n), 256, 0, st>>>(n, (const float*)im, (const float*)offset, g, (float*)columns);
  else
    k_def_im2col<_Float16><<<grid_for(n), 256, 0, st>>>(n, (const _Float16*)im, (const _Float16*)offset, g, (_Float16*)columns);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_deformable_col2im(const void* columns, const void* offset, void* grad_im_acc,
                                     const s2a_dcn_params* p, s2a_stream_t stream) {
  float* grad_im_f32 = (float*)grad_im_acc;      // float32 accumulator (float32 / float16 columns); float64 for float64 columns
  BwdGeom g;
  int rc = make_geom(p, &g, "deformable_col2im");
  if (rc != S2A_OK) return rc;
  const int64_t n = (int64_t)g.C * g.kh * g.kw * g.S * g.Ho * g.Wo;
  if (n == 0) return S2A_OK;
  S2A_CHECK_ARG(columns && offset && grad_im_f32, "deformable_col2im: NULL tensor");
  hipStream_t st = as_stream(stream);
  if (p->dtype == S2A_DTYPE_F64) {
    k_def_col2im<double><<<grid_for(n), 256, 0, st>>>(n, (const double*)columns, (const double*)offset, g, (double*)grad_im_acc);
    S2A_LAUNCH_CHECK();
    return S2A_OK;
  }
  const bool tiled = g.kh == 3 && g.kw == 3 && g.stride_h == 1 && g.stride_w == 1 && g.dil_h == 1 && g.dil_w == 1 &&
                     (g.C / g.dg) % kC2Ch == 0 && !getenv("S2A_COL2IM_SIMPLE");
  if (tiled) {
    dim3 grid((unsigned)((int64_t)g.S * ((g.Ho + kC2TH - 1) / kC2TH) * ((g.Wo + kC2TW - 1) / kC2TW)),
              (unsigned)((g.C + kC2Ch - 1) / kC2Ch));
    if (p->dtype == S2A_DTYPE_F32)
      k_def_col2im_tiled<float><<<grid, 256, 0, st>>>((const float*)columns, (const float*)offset, g, grad_im_f32);
    else
      k_def_col2im_tiled<_Float16><<<grid, 256, 0, st>>>((const _Float16*)columns, (const _Float16*)offset, g, grad_im_f32);
  } else if (p->dtype == S2A_DTYPE_F32)
    k_def_col2im<float><<<grid_for(n), 256, 0, st>>>(n, (const float*)columns, (const float*)offset, g, grad_im_f32);
  else
    k_def_col2im<_Float16><<<grid_for(n), 256, 0, st>>>(n, (const _Float16*)columns, (const _Float16*)offset, g, grad_im_f32);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

extern "C" int s2a_deformable_col2im_coord(const void* columns, const void* im, const void* offset,
                                           void* grad_offset, const s2a_dcn_params* p, s2a_stream_t stream) {
  BwdGeom g;
  int rc = make_geom(p, &g, "deformable_col2im_coord");
  if (rc != S2A_OK) return rc;
  const int64_t n = (int64_t)g.S * 2 * g.kh * g.kw * g.dg * g.Ho * g.Wo;
  if (n == 0) return S2A_OK;
  S2A_CHECK_ARG(columns && im && offset && grad_offset, "deformable_col2im_coord: NULL tensor");
  hipStream_t st = as_stream(stream);
  if (p->dtype == S2A_DTYPE_F64)
    k_def_col2im_coord<double><<<grid_for(n), 256, 0, st>>>(n, (const double*)columns, (const double*)im,
                                                            (const double*)offset, g, (double*)grad_offset);
  else if (p->dtype == S2A_DTYPE_F32)
    k_def_col2im_coord<float><<<grid_for(n), 256, 0, st>>>(n, (const float*)columns, (const float*)im,
                                                           (const float*)offset, g, (float*)grad_offset);
  else
    k_def_col2im_coord<_Float16><<<grid_for(n), 256, 0, st>>>(n, (const _Float16*)columns, (const _Float16*)im,
                                                              (const _Float16*)offset, g, (_Float16*)grad_offset);
  S2A_LAUNCH_CHECK();
  return S2A_OK;
}

// ================================================================= host side of the fused backward (AlignConv geometry)
// deform_conv_backward_input_cuda / deform_conv_backward_parameters_cuda (models/dcn/src/deform_conv_cuda.cpp:262-489) for the
// AlignConv geometry -- 3x3, stride 1, pad 1, dilation 1, one group, one deformable group -- as fused kernels, f16 or f32.
// All tensors NCHW as the reference passes them.  One implementation behind five entry points: the two gradients separately
// (the reference's two functions) or both in one call, which shares the NHWC copies of input and gradOutput between them
// (the autograd backward wants both: two transposes instead of four).
namespace s2a {
namespace {
struct FusedBwdArgs {
  int dtype;                                     // S2A_DTYPE_F16 / S2A_DTYPE_F32
  const void *input, *offset, *grad_output, *weight;
  void* grad_input;                              // [S,C,H,W]; NULL: no input / offset gradient
  int grad_input_typed;                          // 0: f32, ACCUMULATED (+=); 1: of `dtype`, OVERWRITTEN
  void* grad_offset;                             // dtype [S,18,H,W], overwritten
  float* grad_weight;                            // f32 [O,C,3,3], += scale * ...; NULL: no weight gradient
  float scale;
  int64_t B, C, H, W, O;
};

size_t fused_bwd_workspace(int dtype, bool want_input, bool want_weight, int64_t B, int64_t C, int64_t H, int64_t W, int64_t O) {
  const size_t el = dtype == S2A_DTYPE_F16 ? 2 : 4;
  size_t n = align_up((size_t)(B * H * W * C) * el) + align_up((size_t)(B * H * W * O) * el) + 1024;
  if (want_input) n += align_up((size_t)(O * C * 9) * el) + (dtype == S2A_DTYPE_F16 ? align_up((size_t)(B * H * W * C) * 4) : 0);
  if (want_weight) n += align_up((size_t)kWgradMaxBlocks * O * 192 * 4);
  return n;
}

template <typename T>
int fused_bwd_run(const FusedBwdArgs& a, void* workspace, size_t workspace_bytes, hipStream_t st, const char* who) {
  constexpr bool kHalf = sizeof(T) == 2;
  const bool want_input = a.grad_input != nullptr, want_weight = a.grad_weight != nullptr;
  const int64_t B = a.B, C = a.C, H = a.H, W = a.W, O = a.O;
  S2A_CHECK_ARG(B >= 0 && C > 0 && H >= 3 && W >= 3 && O > 0, "%s: bad shape", who);
  S2A_CHECK_ARG(H < (1 << 15) && W < (1 << 15), "%s: shape too large", who);
  S2A_CHECK_ARG(O <= 256, "%s: needs out_channels <= 256", who);
  if (want_input) S2A_CHECK_ARG(C % 32 == 0 && O % 16 == 0, "%s: the input gradient needs channels %% 32 == 0, out_channels %% 16 == 0", who);
  if (want_weight) S2A_CHECK_ARG(C % 64 == 0 && O % 32 == 0, "%s: the weight gradient needs channels %% 64 == 0, out_channels %% 32 == 0", who);
  // (every argument check sits in front of the first launch: a refused call leaves the caller's gradients untouched)
  if (want_weight) S2A_CHECK_ARG(3 * (C / 64) <= kWgradMaxBlocks, "%s: the fused weight gradient takes channels <= %d", who, kWgradMaxBlocks / 3 * 64);
  if (B == 0 || (!want_input && !want_weight)) return S2A_OK;
  S2A_CHECK_ARG(a.input && a.offset && a.grad_output, "%s: NULL tensor", who);
  S2A_CHECK_ARG(!want_input || (a.weight && a.grad_offset), "%s: NULL tensor", who);
  S2A_CHECK_ARG(workspace_bytes >= fused_bwd_workspace(a.dtype, want_input, want_weight, B, C, H, W, O), "%s: workspace too small", who);
  Carver cv(workspace, workspace_bytes);
  const int64_t HW = H * W;
  T* xn = cv.take<T>((size_t)(B * HW * C));
  T* gn = cv.take<T>((size_t)(B * HW * O));
  T* wp = want_input ? cv.take<T>((size_t)(O * C * 9)) : nullptr;
  // f16: [S,H,W,C] f32 accumulator the tiles' atomics add into (128-byte rows); f32: the caller's gradInput itself
  float* gacc = (want_input && kHalf) ? cv.take<float>((size_t)(B * HW * C)) : nullptr;
  float* partial = want_weight ? cv.take<float>((size_t)kWgradMaxBlocks * O * 192) : nullptr;
  S2A_CHECK_ARG(xn && gn && (!want_input || (wp && (gacc || !kHalf))) && (!want_weight || partial), "%s: workspace too small", who);
  // (an own fill kernel, not hipMemsetAsync: memset nodes of a captured graph are not replayed correctly on ROCm 7.2, DESIGN 5 --
  // a backward captured into a HIP graph would otherwise sum into a stale accumulator from its second replay on)
  if (want_input && (kHalf || a.grad_input_typed)) {
    float* z = kHalf ? gacc : (float*)a.grad_input;
    const int64_t nz4 = B * HW * C / 4;          // (C % 32 == 0)
    k_bwd_zero4<<<(unsigned)std::min<int64_t>((nz4 + 255) / 256, 65536), 256, 0, st>>>(reinterpret_cast<f32x4b*>(z), nz4);
  }
  auto to_nhwc = [&](const void* src, int64_t ch, T* dst) {
    if constexpr (kHalf) {
      if (ch % 8 == 0 && HW % 8 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        k_bwd_nchw_to_nhwc_h8<<<dim3((unsigned)((HW + 63) / 64), (unsigned)((ch + 63) / 64), (unsigned)B), 256, 0, st>>>(
            (const _Float16*)src, (int)ch, HW, dst);
        return;
      }
    }
    k_bwd_nchw_to_nhwc<T><<<dim3((unsigned)((HW + 31) / 32), (unsigned)((ch + 31) / 32), (unsigned)B), 256, 0, st>>>((const T*)src, (int)ch, HW, dst);
  };
  to_nhwc(a.input, C, xn);
  to_nhwc(a.grad_output, O, gn);
  if (want_input) {
    const int64_t wtotal = O * C * 9;
    if constexpr (kHalf) {
      k_pack_weight_bwd<<<(unsigned)((wtotal + 255) / 256), 256, 0, st>>>((const _Float16*)a.weight, (int)O, (int)C, wp);
      const int64_t tiles = B * ((H + kBTH - 1) / kBTH) * ((W + kBTW - 1) / kBTW);
      S2A_CHECK_ARG(tiles < (1ll << 31), "%s: too many tiles", who);
      S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dcn_bwd_input), hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLds));
      k_dcn_bwd_input<<<(unsigned)tiles, kBThreads, kBwdLds, st>>>(xn, gn, (const _Float16*)a.offset, wp, gacc, (_Float16*)a.grad_offset,
                                                            (int)B, (int)C, (int)H, (int)W, (int)O);
    } else {
      k_pack_weight_bwd_f32<<<(unsigned)((wtotal + 255) / 256), 256, 0, st>>>((const float*)a.weight, (int)O, (int)C, wp);
      const int64_t tiles = B * ((H + kFTH - 1) / kFTH) * ((W + kFTW - 1) / kFTW);
      S2A_CHECK_ARG(tiles < (1ll << 31), "%s: too many tiles", who);
      const int lds = bwd_f32_lds_bytes((int)O);
      S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dcn_bwd_input_f32<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dcn_bwd_input_f32<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      if (O % 128 == 0)
        k_dcn_bwd_input_f32<true><<<(unsigned)tiles, 512, lds, st>>>(xn, gn, (const float*)a.offset, wp, (float*)a.grad_input, (float*)a.grad_offset,
                                                                    (int)B, (int)C, (int)H, (int)W, (int)O);
      else
        k_dcn_bwd_input_f32<false><<<(unsigned)tiles, 512, lds, st>>>(xn, gn, (const float*)a.offset, wp, (float*)a.grad_input, (float*)a.grad_offset,
                                                                     (int)B, (int)C, (int)H, (int)W, (int)O);
    }
    if constexpr (kHalf) {
      const dim3 fg((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
      if (a.grad_input_typed) k_bwd_acc_to_nchw<T, false><<<fg, 256, 0, st>>>(gacc, (int)C, HW, (T*)a.grad_input);
      else k_bwd_acc_to_nchw<float, true><<<fg, 256, 0, st>>>(gacc, (int)C, HW, (float*)a.grad_input);
    }
    S2A_LAUNCH_CHECK();
  }
  if (want_weight) {
    const int owners = 3 * (int)(C / 64);
    int n_cu = 256;
    {
      int dev = 0;
      hipDeviceProp_t prop;
      S2A_HIP(hipGetDevice(&dev));
      S2A_HIP(hipGetDeviceProperties(&prop, dev));
      if (prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    }
    int ksplit;
    if constexpr (kHalf) {
      const int64_t tiles = B * ((H + kBTH - 1) / kBTH) * ((W + kBTW - 1) / kBTW);
      S2A_CHECK_ARG(tiles < (1ll << 31), "%s: too many tiles", who);
      ksplit = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, std::min(n_cu, kWgradMaxBlocks) / owners));
      const int dw = (int)O / 2;
      const int gop = (dw + ((16 - (dw & 63)) & 63)) * 4;
      const int lds = kWPos * gop + kWPatchPix * 128 + 3 * kWPos * kWColRow + 3 * kWPos * 16;
      S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dcn_bwd_weight), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      k_dcn_bwd_weight<<<(unsigned)(owners * ksplit), 512, lds, st>>>(xn, gn, (const _Float16*)a.offset, partial, (int)B, (int)C, (int)H,
                                                                   (int)W, (int)O, ksplit);
    } else {
      const int64_t tiles = B * ((H + kFTH - 1) / kFTH) * ((W + kFTW - 1) / kFTW);
      S2A_CHECK_ARG(tiles < (1ll << 31), "%s: too many tiles", who);
      ksplit = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, std::min(n_cu, kWgradMaxBlocks) / owners));
      // default: the f32 tensors as three bf16 planes on the 16-bit matrix instruction (k_dcn_bwd_weight_x3); S2A_BWD_F32_WEIGHT=mfma32:
      // the f32 instruction's kernel
      const char* wsel = std::getenv("S2A_BWD_F32_WEIGHT");
      if (wsel && wsel[0] == 'm') {
        const int lds = wgrad_f32_lds_bytes();
        S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dcn_bwd_weight_f32), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        k_dcn_bwd_weight_f32<<<(unsigned)((owners * ksplit + 7) / 8 * 8), kFWThreads, lds, st>>>(xn, gn, (const float*)a.offset, partial, (int)B,
                                                                                               (int)C, (int)H, (int)W, (int)O, ksplit);
      } else {
        const int lds = wgrad_x3_lds_bytes((int)O);
        S2A_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dcn_bwd_weight_x3), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        k_dcn_bwd_weight_x3<<<(unsigned)(owners * ksplit), 512, lds, st>>>(xn, gn, (const float*)a.offset, partial, (int)B, (int)C, (int)H,
                                                                         (int)W, (int)O, ksplit);
      }
    }
    const int64_t nout = O * C * 9;
    k_dcn_bwd_weight_reduce<<<(unsigned)((nout + 255) / 256), 256, 0, st>>>(partial, a.grad_weight, a.scale, (int)O, (int)C, ksplit);
    S2A_LAUNCH_CHECK();
  }
  return S2A_OK;
}

int fused_bwd(const FusedBwdArgs& a, void* workspace, size_t workspace_bytes, s2a_stream_t stream, const char* who) {
  S2A_CHECK_ARG(a.dtype == S2A_DTYPE_F16 || a.dtype == S2A_DTYPE_F32, "%s: float16 / float32", who);
  return a.dtype == S2A_DTYPE_F16 ? fused_bwd_run<_Float16>(a, workspace, workspace_bytes, as_stream(stream), who)
                                  : fused_bwd_run<float>(a, workspace, workspace_bytes, as_stream(stream), who);
}
}  // namespace
}  // namespace s2a

// f16: grad_input_f32 [S,C,H,W] is ACCUMULATED (the caller zeroes it), grad_offset [S,18,H,W] f16 is overwritten
extern "C" size_t s2a_deform_conv_backward_input_workspace_bytes(int64_t batch, int64_t channels, int64_t height,
                                                                 int64_t width, int64_t out_channels) {
  return fused_bwd_workspace(S2A_DTYPE_F16, true, false, batch, channels, height, width, out_channels);
}
extern "C" int s2a_deform_conv_backward_input_f16(const void* input, const void* offset, const void* grad_output,
                                                  const void* weight, float* grad_input_f32, void* grad_offset,
                                                  int64_t batch, int64_t channels, int64_t height, int64_t width,
                                                  int64_t out_channels, void* workspace, size_t workspace_bytes,
                                                  s2a_stream_t stream) {
  S2A_CHECK_ARG(batch == 0 || grad_input_f32, "deform_conv_backward_input_f16: NULL tensor");
  return fused_bwd(FusedBwdArgs{S2A_DTYPE_F16, input, offset, grad_output, weight, grad_input_f32, 0, grad_offset, nullptr, 1.f, batch,
                                channels, height, width, out_channels},
                   workspace, workspace_bytes, stream, "deform_conv_backward_input_f16");
}

// f32: grad_input [S,C,H,W] is the caller's gradInput, ACCUMULATED in place; grad_offset [S,18,H,W] f32 is overwritten
extern "C" size_t s2a_deform_conv_backward_input_f32_workspace_bytes(int64_t batch, int64_t channels, int64_t height,
                                                                     int64_t width, int64_t out_channels) {
  return fused_bwd_workspace(S2A_DTYPE_F32, true, false, batch, channels, height, width, out_channels);
}
extern "C" int s2a_deform_conv_backward_input_f32(const float* input, const float* offset, const float* grad_output,
                                                  const float* weight, float* grad_input, float* grad_offset,
                                                  int64_t batch, int64_t channels, int64_t height, int64_t width,
                                                  int64_t out_channels, void* workspace, size_t workspace_bytes,
                                                  s2a_stream_t stream) {
  S2A_CHECK_ARG(batch == 0 || grad_input, "deform_conv_backward_input_f32: NULL tensor");
  return fused_bwd(FusedBwdArgs{S2A_DTYPE_F32, input, offset, grad_output, weight, grad_input, 0, grad_offset, nullptr, 1.f, batch, channels,
                                height, width, out_channels},
                   workspace, workspace_bytes, stream, "deform_conv_backward_input_f32");
}

// f16: grad_weight_f32 [O,C,3,3] is ACCUMULATED, unscaled (the caller zeroes it and applies `scale`)
extern "C" size_t s2a_deform_conv_backward_weight_workspace_bytes(int64_t batch, int64_t channels, int64_t height,
                                                                  int64_t width, int64_t out_channels) {
  return fused_bwd_workspace(S2A_DTYPE_F16, false, true, batch, channels, height, width, out_channels);
}
extern "C" int s2a_deform_conv_backward_weight_f16(const void* input, const void* offset, const void* grad_output,
                                                   float* grad_weight_f32, int64_t batch, int64_t channels, int64_t height,
                                                   int64_t width, int64_t out_channels, void* workspace,
                                                   size_t workspace_bytes, s2a_stream_t stream) {
  S2A_CHECK_ARG(batch == 0 || grad_weight_f32, "deform_conv_backward_weight_f16: NULL tensor");
  return fused_bwd(FusedBwdArgs{S2A_DTYPE_F16, input, offset, grad_output, nullptr, nullptr, 0, nullptr, grad_weight_f32, 1.f, batch,
                                channels, height, width, out_channels},
                   workspace, workspace_bytes, stream, "deform_conv_backward_weight_f16");
}

// f32: grad_weight [O,C,3,3] is the caller's gradWeight, += scale * gradOutput x columns^T in place
extern "C" size_t s2a_deform_conv_backward_weight_f32_workspace_bytes(int64_t batch, int64_t channels, int64_t height,
                                                                      int64_t width, int64_t out_channels) {
  return fused_bwd_workspace(S2A_DTYPE_F32, false, true, batch, channels, height, width, out_channels);
}
extern "C" int s2a_deform_conv_backward_weight_f32(const float* input, const float* offset, const float* grad_output,
                                                   float* grad_weight, float scale, int64_t batch, int64_t channels,
                                                   int64_t height, int64_t width, int64_t out_channels, void* workspace,
                                                   size_t workspace_bytes, s2a_stream_t stream) {
  S2A_CHECK_ARG(batch == 0 || grad_weight, "deform_conv_backward_weight_f32: NULL tensor");
  return fused_bwd(FusedBwdArgs{S2A_DTYPE_F32, input, offset, grad_output, nullptr, nullptr, 0, nullptr, grad_weight, scale, batch, channels,
                                height, width, out_channels},
                   workspace, workspace_bytes, stream, "deform_conv_backward_weight_f32");
}

// Both gradients in one call (what DeformConvFunction.backward needs, deform_conv.py:73-118): the NHWC copies of input and
// gradOutput are made once.  dtype = S2A_DTYPE_F16 / S2A_DTYPE_F32 is the type of input, offset, grad_output, weight and
// grad_offset; grad_input (f32 [S,C,H,W], accumulated) and grad_weight (f32 [O,C,3,3], += scale * ...) are always f32;
// either of the two may be NULL to skip that gradient (grad_offset goes with grad_input).
extern "C" size_t s2a_deform_conv_backward_workspace_bytes(int dtype, int64_t batch, int64_t channels, int64_t height,
                                                           int64_t width, int64_t out_channels) {
  return fused_bwd_workspace(dtype, true, true, batch, channels, height, width, out_channels);
}
extern "C" int s2a_deform_conv_backward(int dtype, const void* input, const void* offset, const void* grad_output,
                                        const void* weight, float* grad_input_f32, void* grad_offset, float* grad_weight_f32,
                                        float scale, int64_t batch, int64_t channels, int64_t height, int64_t width,
                                        int64_t out_channels, void* workspace, size_t workspace_bytes, s2a_stream_t stream) {
  return fused_bwd(FusedBwdArgs{dtype, input, offset, grad_output, weight, grad_input_f32, 0, grad_offset, grad_weight_f32, scale, batch,
                                channels, height, width, out_channels},
                   workspace, workspace_bytes, stream, "deform_conv_backward");
}
// The same call with gradInput as DeformConvFunction.backward hands it over (deform_conv.py:88: zeros_like(input)): a tensor of
// `dtype`, [S,C,H,W], OVERWRITTEN with the gradient (no f32 copy on the caller's side, no conversion pass behind the call).
// Workspace: s2a_deform_conv_backward_workspace_bytes.
extern "C" int s2a_deform_conv_backward_typed(int dtype, const void* input, const void* offset, const void* grad_output,
                                              const void* weight, void* grad_input, void* grad_offset, float* grad_weight_f32,
                                              float scale, int64_t batch, int64_t channels, int64_t height, int64_t width,
                                              int64_t out_channels, void* workspace, size_t workspace_bytes, s2a_stream_t stream) {
  return fused_bwd(FusedBwdArgs{dtype, input, offset, grad_output, weight, grad_input, 1, grad_offset, grad_weight_f32, scale, batch,
                                channels, height, width, out_channels},
                   workspace, workspace_bytes, stream, "deform_conv_backward_typed");
}
